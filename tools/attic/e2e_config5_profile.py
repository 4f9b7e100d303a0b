#!/usr/bin/env python3
"""cProfile (main thread) of the two calls of config 5, one after the other:
`python tools/e2e_config5_profile.py --samples 4 --reads 20000000`."""
import argparse
import cProfile
import contextlib
import io
import os
import pstats
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=2)
    ap.add_argument('--reads', type=int, default=1_000_000)
    ap.add_argument('--dir', default=None)
    ap.add_argument('--top', type=int, default=45)
    ap.add_argument('--plain', action='store_true', help='no cProfile: wall times only, twice')
    a = ap.parse_args()
    import bench
    from woltka_amd import workflow
    with tempfile.TemporaryDirectory(dir=a.dir) as tmp:
        fps, n_rec, n_bytes, one = bench.write_twopass_inputs(tmp, a.samples, a.reads)
        kw1, kw2 = bench.twopass_calls(fps, tmp)
        for rep in range(2 if a.plain else 1):
            shutil.rmtree(kw1['outmap_dir'], ignore_errors=True)
            for key, kw in (('pass1', kw1), ('pass2', kw2)):
                pr = cProfile.Profile()
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    if a.plain:
                        workflow.workflow(**kw)
                    else:
                        pr.runcall(workflow.workflow, **kw)
                dt = time.perf_counter() - t0
                print(f'== {key}: {dt:.3f} s, {n_rec / dt / 1e6:.1f} M records/s', flush=True)
                if not a.plain:
                    pstats.Stats(pr).sort_stats('cumulative').print_stats(a.top)
                    pstats.Stats(pr).sort_stats('tottime').print_stats(25)


if __name__ == '__main__':
    main()
