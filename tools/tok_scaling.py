#!/usr/bin/env python3
"""Host-side scaling of the alignment tokenizer on this box: a config-3-shaped
SAM file (page cache / tmpfs) through `align.native_sam_blocks` — the driver
`woltka classify` uses, small ramp blocks first — at several thread counts.
No GPU involved.  WOLTKA_TOK_TIMING=1 adds the per-phase totals.

    python tools/tok_scaling.py [--reads 25000000] [--threads 8,16,32,64,128]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402
from woltka_amd import align, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=25_000_000)
    ap.add_argument('--threads', default='8,16,32,64,128')
    ap.add_argument('--block', type=int, default=1 << 28)
    ap.add_argument('--dir', default='/dev/shm' if os.path.isdir('/dev/shm')
                    else '/tmp')
    ap.add_argument('--packed', action='store_true')
    a = ap.parse_args()
    fp = os.path.join(a.dir, f'tok_scaling_{a.reads}.sam')
    if not os.path.exists(fp):
        rng = np.random.default_rng(1003)
        p = synth.as_sets(synth.lca_problem(
            rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=a.reads,
            with_names=False))
        t0 = time.perf_counter()
        n_rec, size = bench.write_sam_lca(fp, p, a.reads)
        print(f'wrote {n_rec} records, {size / 1e9:.2f} GB in '
              f'{time.perf_counter() - t0:.1f} s', flush=True)
        del p
    size = os.path.getsize(fp)
    for threads in [int(x) for x in a.threads.split(',')]:
        for rep in range(2):
            tok = nat.Tokenizer(threads)
            n_rec = 0
            t0 = time.perf_counter()
            with open(fp, 'rb') as f:
                for buf, res in align.native_sam_blocks(f, tok, a.block):
                    n_rec += res['subj'].size
                    tok.new_subjects()
            dt = time.perf_counter() - t0
            print(f'threads {threads:4d} run {rep}: {n_rec / dt / 1e6:8.1f} M '
                  f'records/s  {size / dt / 1e9:6.2f} GB/s  {dt:.3f} s',
                  flush=True)
            tok.close()


if __name__ == '__main__':
    main()
