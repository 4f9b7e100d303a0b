#!/usr/bin/env python3
"""End-to-end throughput: synthetic SAM text on local disk -> table.

Generates config-2-shaped SAM (1 hit per read, trimmed lines as recommended in
the reference's doc/perform.md:122-128), then times `workflow.classify` through
the native tokenizer + GPU path, and the tokenizer alone.
"""
import argparse
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from woltka_amd import _native as nat  # noqa: E402
from woltka_amd import align, workflow  # noqa: E402
from woltka_amd.synth import zipf_draw  # noqa: E402


def make_sam(path, n_reads, n_subjects=10575, seed=1002):
    rng = np.random.default_rng(seed)
    sub = zipf_draw(rng, n_subjects, n_reads)
    with open(path, 'wb') as f:
        f.write(b'@HD\tVN:1.0\tSO:unsorted\n')
        step = 1_000_000
        for lo in range(0, n_reads, step):
            hi = min(n_reads, lo + step)
            lines = [b'R%09d\t0\tG%09d\t%d\t42\t150M\t*\t0\t0\t*\t*\n' %
                     (i, s, 1 + (i * 7919) % 4000000)
                     for i, s in zip(range(lo, hi), sub[lo:hi].tolist())]
            f.write(b''.join(lines))
    return os.path.getsize(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=10_000_000)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--dir', default=tempfile.gettempdir())
    ap.add_argument('--no-gpu', action='store_true')
    a = ap.parse_args()
    fp = os.path.join(a.dir, f'synth_{a.reads}.sam')
    t0 = time.perf_counter()
    size = make_sam(fp, a.reads) if not os.path.isfile(fp) else os.path.getsize(fp)
    print(f'SAM: {a.reads} records, {size / 1e6:.1f} MB '
          f'({size / a.reads:.1f} B/record), generated in {time.perf_counter() - t0:.1f} s')
    # tokenizer alone (file already in the page cache)
    for threads in ([a.threads] if a.threads else [1, 8, 32, 0]):
        tok = nat.Tokenizer(threads)
        t0 = time.perf_counter()
        nrec = 0
        with open(fp, 'rb') as f:
            for buf, res in align.native_sam_blocks(f, tok, 1 << 27):
                nrec += res['subj'].size
                del buf, res
        dt = time.perf_counter() - t0
        print(f'tokenizer threads={threads or os.cpu_count()}: {nrec / dt / 1e6:.2f} M records/s '
              f'({size / dt / 1e9:.2f} GB/s text) in {dt:.2f} s')
        tok.close()
    if a.no_gpu:
        return
    import click
    for rep in range(2):
        t0 = time.perf_counter()
        with open(os.devnull, 'w') as devnull:
            import contextlib
            with contextlib.redirect_stdout(devnull):
                data = workflow.classify(align.plain_mapper, {fp: 'S1'},
                                         ['S1'], fmt='sam', ranks=['none'])
        dt = time.perf_counter() - t0
        tot = sum(data['none']['S1'].values())
        print(f'end-to-end classify (rank none): {a.reads / dt / 1e6:.2f} M records/s '
              f'in {dt:.2f} s; counted {tot}')


if __name__ == '__main__':
    main()
