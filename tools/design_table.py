#!/usr/bin/env python3
"""The results table of DESIGN.md §5 from a bench.py line:

    python tools/design_table.py profiles/r05_bench_final.json
"""
import json
import sys

d = json.load(open(sys.argv[1]))


def g(x, *keys, default=None):
    for k in keys:
        if not isinstance(x, dict) or k not in x:
            return default
        x = x[k]
    return x


def rate(v):
    if v is None:
        return '—'
    if v < 1e6:
        return f'{v / 1e6:.2f} M'
    return f'{v / 1e6:.0f} M' if v < 1e10 else (
        f'{v / 1e9:.1f} G' if v < 1e11 else f'{v / 1e9:.0f} G')


rows = []
r = d['roofline']
rows.append(('**headline** (`value`): device side of R1, config 3, text resident in HBM',
             f"{d['config']['ms_per_pass']:.1f} ms per pass over 10.5 GB = **{rate(d['value'])} records/s**",
             f"dominant family `{r['kernel']}` {r['kernel_ms'] * 1e3:.0f} us per 64 MB block = {r['achieved']:.0f} GB/s = **{r['frac']:.3f}** of the HBM peak by text + words; "
             f"families per block: " + ', '.join(f"{k} {v * 1e3:.0f} us" for k, v in r['kernels_ms'].items())))
for key, label in (('lca', 'histogram alone (`configs.lca`, resident sliced records)'),
                   ('lca_free', '`--rank free`'), ('lca_above', '`--above`'),
                   ('lca_major', '`--major 80`'), ('lca_uniq', '`--uniq`'),
                   ('lca_above3', '`--rank phylum,genus,species --above` (one route, a stream pass per rank)'),
                   ('ordinal', 'coord-match (`configs.ordinal`, 107.5 M hits staged)'),
                   ('flat', 'config 2 (`configs.flat`, 8 x 10 M records)')):
    c = g(d, 'configs', key)
    if not c or 'error' in c:
        continue
    rf, rs = c['roofline'], c['roofline_step']
    extra = ''
    if 'sort_ms' in c:
        extra = (f"; stripe sort once per staged chunk {c['sort_ms']:.2f} ms "
                 f"(pass + sort {c['ms_per_pass_with_sort']:.2f} ms, "
                 f"{c['roofline_step_with_sort']['frac']:.3f})")
    rows.append((label, f"{c['ms_per_pass']:.3f} ms per pass = {rate(c['value'])} records/s",
                 f"`{rf['kernel']}` {rf['kernel_ms']:.3f} ms = {rf['frac']:.3f}; whole step {rs['frac']:.3f}{extra}"))
for key, label in (('lca', 'config 3, `woltka classify` end to end'),
                   ('lca_gz', 'the same text as one `.sam.gz`'),
                   ('lca_gz8', 'the same text as eight `.sam.gz`'),
                   ('lca_seqqual', 'config 3 with 150-base SEQ / QUAL on every line (85 GB)'),
                   ('flat', 'config 2 end to end (10 M records, flat map)'),
                   ('ordinal', 'config 4 end to end')):
    e = g(d, 'e2e', key)
    if not e or 'error' in e:
        continue
    rf = e.get('roofline', {})
    if rf.get('bound') == 'host_scan':
        bound = (f"{rf.get('achieved', '—')} GB/s of file text, above the link's "
                 f"{rf.get('peak', '—')} GB/s: the columns nobody reads are cut on the host (bound: the host's scan)")
    else:
        bound = f"{rf.get('achieved', '—')} of {rf.get('peak', '—')} GB/s measured H2D = {rf.get('frac', '—')}"
    rows.append((f'`e2e.{key}`: {label}',
                 f"{e['seconds']:.3f} s = **{rate(e['value'])} records/s**",
                 f"{bound}; phases {e.get('phases_s')}"))
tp = g(d, 'e2e', 'twopass')
if tp and 'error' not in tp:
    for k in ('pass1', 'pass2'):
        e = tp[k]
        rf = e.get('roofline', {})
        rows.append((f'`e2e.twopass.{k}`: config 5 on one GPU (8 samples x 20 M reads, 37 GB)',
                     f"{e['seconds']:.3f} s = **{rate(e['value'])} records/s**",
                     f"{rf.get('achieved', '—')} of {rf.get('peak', '—')} GB/s = {rf.get('frac', '—')}; phases {e.get('phases_s')}"))
cb = d.get('cpu_baseline', {})
if cb and 'value' in cb:
    rows.append(('`cpu_baseline`: pure-Python restatement, one core', f"{rate(cb['value'])} records/s", cb.get('sample', '')[:120]))
print('| What | Time / rate | Roofline |')
print('|---|---|---|')
for a, b, c in rows:
    print(f'| {a} | {b} | {c} |')
