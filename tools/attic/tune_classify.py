#!/usr/bin/env python3
"""Sweep the classify kernel's launch knobs on a bench workload and print the
HIP-event kernel time per configuration (GPU box only)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='flat')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--ablate', type=int, default=0)
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--tiled', type=int, default=-1)
    ap.add_argument('--dense', type=int, default=1)
    a = ap.parse_args()
    ctx = nat.Context(0)
    wl = bench.WORKLOADS[a.workload](ctx, 1002, a.scale)
    ctx.profile_kernels(True)
    ctx.tune('ablate', a.ablate)
    ctx.tune('dense', a.dense)
    print(f'workload {wl.name}: {wl.records} records, {wl.alg_bytes / 1e6:.1f} MB algorithmic')
    cfgs = []
    for tiled in ((1, 0) if a.tiled < 0 else (a.tiled,)):
        for slots in ((4096,) if a.quick else (1024, 2048, 4096, 8192)):
            for threads, bpc in (((512, 2),) if tiled else ((1024, 1), (1024, 2), (512, 2), (512, 4), (512, 3), (256, 8))):
                if tiled and slots > 4096:
                    continue
                cfgs.append((tiled, slots, threads, bpc))
    for tiled, slots, threads, bpc in cfgs:
            if True:
                ctx.tune('tiled', tiled)
                ctx.tune('lds_slots', slots)
                ctx.tune('threads', threads)
                ctx.tune('blocks_per_cu', bpc)
                ctx.counts_clear()
                ts = []
                for _ in range(a.reps):
                    wl.step()
                    ts.append(ctx.last_kernel_ms('classify'))
                ms = float(np.median(ts))
                print(f'tiled={tiled} slots={slots:5d} thr={threads:4d} blocks/CU={bpc:2d}  '
                      f'{ms * 1e3:9.1f} us  {wl.alg_bytes / ms / 1e6:8.1f} GB/s  '
                      f'{wl.records / ms / 1e6:8.2f} Grec/s', flush=True)

if __name__ == '__main__':
    main()
