#!/usr/bin/env python3
"""BASELINE configs[4] (two-pass stratified workflow) at any size, through
bench.e2e_twopass: `python tools/e2e_config5.py --samples 8 --reads 20000000`."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=2)
    ap.add_argument('--reads', type=int, default=1_000_000)
    ap.add_argument('--reps', type=int, default=1)
    ap.add_argument('--dir', default=None)
    ap.add_argument('--digest', action='store_true')
    a = ap.parse_args()
    import bench
    res = bench.e2e_twopass(0, a.samples, a.reads, workdir=a.dir, reps=a.reps,
                            digest=a.digest)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
