#!/usr/bin/env python3
"""cProfile of one warm `woltka classify` call of an end-to-end leg
(tools/e2e_once.py kinds): where the host's time goes.

    python tools/e2e_cprofile.py <kind> [--dir D] [--reads N] [--top 40]
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
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('kind')
    ap.add_argument('--dir', default='/dev/shm/wk_e2e_prof')
    ap.add_argument('--reads', type=int, default=0)
    ap.add_argument('--top', type=int, default=40)
    a = ap.parse_args()
    import bench
    from woltka_amd import workflow
    os.makedirs(a.dir, exist_ok=True)
    meta = bench.e2e_inputs(a.kind, a.dir, a.reads)
    kw = meta['kwargs']

    def run():
        out = kw['output_fp']
        if os.path.isdir(out):
            shutil.rmtree(out)
        with contextlib.redirect_stdout(io.StringIO()):
            workflow.workflow(**kw)
    for rep in range(2):
        t0 = time.perf_counter()
        run()
        print(f'run {rep}: {time.perf_counter() - t0:.3f} s', flush=True)
    os.environ['WOLTKA_DTOK_TIMING'] = '1'
    t0 = time.perf_counter()
    run()
    print(f'timed run: {time.perf_counter() - t0:.3f} s', flush=True)
    del os.environ['WOLTKA_DTOK_TIMING']
    pr = cProfile.Profile()
    pr.enable()
    run()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(a.top)
    print(s.getvalue())
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == '__main__':
    main()
