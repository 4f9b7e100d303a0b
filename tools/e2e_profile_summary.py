#!/usr/bin/env python3
"""What tools/prof_e2e.sh collected, as one JSON: per kernel of the traced
`woltka classify` call its launches, time and (from the PMC passes) HBM bytes;
the call's totals; the device-side roofline of the text route.

    python tools/e2e_profile_summary.py gpurun_out/<tag> <kind> <reps>

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (the gfx950
correction of MI355X_MICROARCH.md's HBM section), the two counters come from
separate passes."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out, kind, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
sys.path.insert(0, root)


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void ', '', name)
    return name.replace('wk::', '')


kern = {}
fp = os.path.join(out, 'kernel_stats.csv')
if os.path.isfile(fp):
    for row in csv.DictReader(open(fp)):
        kern[short(row['Name'])] = {
            'calls': int(row['Calls']),
            'total_ms': round(float(row['TotalDurationNs']) / 1e6, 3),
            'avg_us': round(float(row['AverageNs']) / 1e3, 2)}
pmc = defaultdict(lambda: defaultdict(float))
for fp in glob.glob(os.path.join(out, 'pmc*.csv')):
    for row in csv.DictReader(open(fp)):
        if row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            pmc[short(row['Kernel_Name'])][row['Counter_Name']] += \
                float(row['Counter_Value'])
for name, c in pmc.items():
    k = kern.setdefault(name, {})
    k['hbm_read_bytes'] = int(2 * c.get('FETCH_SIZE', 0.0) * 1024)
    k['hbm_write_bytes'] = int(c.get('WRITE_SIZE', 0.0) * 1024)
runs = []
for log in ('plain.log', 'kt.log'):
    try:
        with open(os.path.join(out, log)) as f:
            for line in f:
                if line.startswith('{"kind"'):
                    runs.append((log, json.loads(line)))
    except OSError:
        pass
meta = runs[0][1] if runs else {}
total_ms = sum(k.get('total_ms', 0.0) for k in kern.values())
hbm = sum(k.get('hbm_read_bytes', 0) + k.get('hbm_write_bytes', 0)
          for k in kern.values())
res = {'kind': kind, 'calls_traced': reps,
       'records': meta.get('records'), 'text_bytes': meta.get('text_bytes'),
       'seconds_plain': dict(runs).get('plain.log', {}).get('seconds'),
       'seconds_traced': dict(runs).get('kt.log', {}).get('seconds'),
       'kernels_ms_per_call': round(total_ms / reps, 3),
       'hbm_bytes_per_call': hbm // reps if hbm else None,
       'kernels': dict(sorted(kern.items(),
                              key=lambda kv: -kv[1].get('total_ms', 0.0)))}
if meta.get('text_bytes') and total_ms:
    # the device side of the text route: the text is the algorithmic input,
    # every kernel of the call counts
    gbs = meta['text_bytes'] / (total_ms / reps * 1e-3) / 1e9
    res['device_roofline'] = {
        'bound': 'hbm', 'bytes': meta['text_bytes'],
        'kernels_ms_sum': round(total_ms / reps, 3),
        'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s',
        'frac': round(gbs / 8000.0, 4),
        'traffic': hbm // reps if hbm else None}
try:
    from woltka_amd import _native as nat
    import __graft_entry__ as ge
    res['build_id'] = nat.build_id()
    res['device_digest'] = ge.device_digest()
except Exception as e:      # noqa: BLE001
    res['build_id'] = repr(e)
with open(os.path.join(out, 'e2e_profile.json'), 'w') as f:
    json.dump(res, f, indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'kernels'}))
for name, k in list(res['kernels'].items())[:16]:
    print(f'  {name[:58]:58s} {k.get("calls", 0):6d} x {k.get("avg_us", 0):9.1f} us'
          f' = {k.get("total_ms", 0):9.2f} ms'
          f'  rd {k.get("hbm_read_bytes", 0) / 1e9:7.2f} GB wr '
          f'{k.get("hbm_write_bytes", 0) / 1e9:7.2f} GB')
