#!/usr/bin/env python3
"""Host-side scaling of the alignment tokenizer on this box: config-3-shaped
SAM files (tmpfs / page cache) through `align.native_sam_blocks` — the driver
`woltka classify` uses, small ramp blocks first — at several thread counts,
in one process or in several at once (each with a file of its own, like the
ranks of a node).  No GPU involved.  WOLTKA_TOK_TIMING=1 adds per-phase totals.

    python tools/tok_scaling.py [--reads 25000000] [--threads 8,16,32,64]
                                [--procs 1,2,4,8] [--mode pread|mmap]
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_file(fp, reads):
    import bench
    from woltka_amd import synth
    rng = np.random.default_rng(1003)
    p = synth.as_sets(synth.lca_problem(
        rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=reads,
        with_names=False))
    t0 = time.perf_counter()
    n_rec, size = bench.write_sam_lca(fp, p, reads)
    print(f'wrote {n_rec} records, {size / 1e9:.2f} GB in '
          f'{time.perf_counter() - t0:.1f} s', flush=True)


def one_pass(fp, threads, block, start_at=None):
    from woltka_amd import _native as nat
    from woltka_amd import align
    tok = nat.Tokenizer(threads)
    if start_at:
        while time.time() < start_at:
            time.sleep(0.001)
    n_rec = 0
    t0 = time.perf_counter()
    with open(fp, 'rb') as f:
        for buf, res in align.native_sam_blocks(f, tok, block):
            n_rec += res['subj'].size
            tok.new_subjects()
    dt = time.perf_counter() - t0
    tok.close()
    return n_rec, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=25_000_000)
    ap.add_argument('--threads', default='8,16,32,64')
    ap.add_argument('--procs', default='1')
    ap.add_argument('--block', type=int, default=1 << 28)
    ap.add_argument('--dir', default='/dev/shm' if os.path.isdir('/dev/shm')
                    else '/tmp')
    ap.add_argument('--mode', default='pread')
    ap.add_argument('--worker', default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    os.environ['WOLTKA_READ'] = a.mode
    if a.worker:
        fp, threads, start_at = a.worker.split(',')
        n_rec, dt = one_pass(fp, int(threads), a.block, float(start_at))
        print(f'{n_rec} {dt:.4f}')
        return
    base = os.path.join(a.dir, f'tok_scaling_{a.reads}')
    max_procs = max(int(x) for x in a.procs.split(','))
    files = [f'{base}_{i}.sam' for i in range(max_procs)]
    if not os.path.exists(files[0]):
        make_file(files[0], a.reads)
    import shutil
    for fp in files[1:]:
        if not os.path.exists(fp):
            shutil.copyfile(files[0], fp)
    size = os.path.getsize(files[0])
    for procs in [int(x) for x in a.procs.split(',')]:
        for threads in [int(x) for x in a.threads.split(',')]:
            if procs == 1:
                for rep in range(2):
                    n_rec, dt = one_pass(files[0], threads, a.block)
                    print(f'{a.mode:5s} procs 1 threads {threads:4d} run {rep}: '
                          f'{n_rec / dt / 1e6:8.1f} M records/s  '
                          f'{size / dt / 1e9:6.2f} GB/s  {dt:.3f} s', flush=True)
                continue
            start_at = time.time() + 3.0
            ps = [subprocess.Popen(
                [sys.executable, __file__, '--mode', a.mode, '--block',
                 str(a.block), '--worker', f'{files[i]},{threads},{start_at}'],
                stdout=subprocess.PIPE, text=True) for i in range(procs)]
            outs = [p.communicate()[0].split() for p in ps]
            tot = sum(int(o[0]) for o in outs)
            wall = max(float(o[1]) for o in outs)
            print(f'{a.mode:5s} procs {procs} threads {threads:4d} each: '
                  f'{tot / wall / 1e6:8.1f} M records/s aggregate  '
                  f'{procs * size / wall / 1e9:6.2f} GB/s  slowest {wall:.3f} s  '
                  f'(per process {tot / procs / wall / 1e6:.1f} M)', flush=True)
    for fp in files:
        try:
            os.remove(fp)
        except OSError:
            pass


if __name__ == '__main__':
    main()
