#!/usr/bin/env python3
"""A `woltka classify` call as a process of its own (what bench.py's
cold_process_s times), with the route's timing notes on stderr:
    python tools/e2e_once.py lca --dir D --prepare; python tools/cold_call.py lca --dir D
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('kind')
ap.add_argument('--dir', required=True)
ap.add_argument('--reps', type=int, default=2)
a = ap.parse_args()
with open(os.path.join(a.dir, f'{a.kind}.meta.json')) as f:
    kw = json.load(f)['kwargs']
flag = {'input_fp': '--input', 'output_fp': '--output', 'input_fmt':
        '--format', 'ranks': '--rank', 'coords_fp': '--coords', 'overlap':
        '--overlap', 'strata_dir': '--stratify', 'outmap_dir': '--outmap'}
many = {'nodes_fps': '--nodes', 'map_fps': '--map'}
INNER = ("import sys, time; t0 = time.perf_counter(); "
         "from woltka_amd.cli import cli; t1 = time.perf_counter(); "
         "cli(sys.argv[1:], standalone_mode=False); t2 = time.perf_counter(); "
         "print('[cold] imports %.3f s, call %.3f s' % (t1 - t0, t2 - t1), "
         "file=sys.stderr); "
         "import os; os.environ.get('COLD_HARD_EXIT') and (sys.stderr.flush(), os._exit(0))")
cmd = [sys.executable, '-c', INNER, 'classify']
for k, v in kw.items():
    if k in flag:
        cmd += [flag[k], str(v)]
    elif k in many:
        for x in v:
            cmd += [many[k], x]
    elif k == 'output_fmt' and v is False:
        cmd.append('--to-tsv')
env = dict(os.environ, PYTHONPATH=ROOT, WOLTKA_DTOK_TIMING='1')
for _ in range(a.reps):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE, text=True, stdin=subprocess.DEVNULL)
    print('cold call: %.3f s, rc %d' % (time.perf_counter() - t0, p.returncode))
    print(p.stderr[-3000:])
