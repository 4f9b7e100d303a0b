#!/usr/bin/env python3
"""Where the HOST time of one `woltka classify` call goes: cProfile around the
second of two `workflow.workflow` calls on inputs `tools/e2e_once.py --prepare`
made (the first call warms caches and the device).

    python tools/prof_host.py twopass2 --dir /dev/shm/wk_e2e [--top 30]
"""
import argparse
import contextlib
import cProfile
import io
import json
import os
import pstats
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('kind')
    ap.add_argument('--dir', required=True)
    ap.add_argument('--top', type=int, default=30)
    a = ap.parse_args()
    from woltka_amd import workflow
    import bench
    with open(os.path.join(a.dir, f'{a.kind}.meta.json')) as f:
        kw = json.load(f)['kwargs']

    def call():
        out = kw['output_fp']
        if os.path.isdir(out):
            shutil.rmtree(out)
        if kw.get('outmap_dir'):
            shutil.rmtree(kw['outmap_dir'], ignore_errors=True)
        bench.wait_closed()
        with contextlib.redirect_stdout(io.StringIO()):
            workflow.workflow(device=0, **kw)
    call()
    pr = cProfile.Profile()
    pr.enable()
    call()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(a.top)
    st.sort_stats('tottime').print_stats(a.top)


if __name__ == '__main__':
    main()
