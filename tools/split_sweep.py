#!/usr/bin/env python3
"""Step time of a bench workload under the two-class split options."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='flat')
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--steps', type=int, default=30)
a = ap.parse_args()
ctx = nat.Context(0)
wl = bench.WORKLOADS[a.workload](ctx, 1002, a.scale)
configs = [dict(split=0), dict(split=2, single_blocks_per_cu=1),
           dict(split=2, single_blocks_per_cu=2),
           dict(split=2, single_blocks_per_cu=2, threads=512),
           dict(split=2, single_blocks_per_cu=4, threads=512),
           dict(split=2, single_blocks_per_cu=4, threads=256),
           dict(split=2, single_blocks_per_cu=8, threads=256)]
for cfg in configs:
    ctx.tune('threads', 1024)
    for k, v in cfg.items():
        ctx.tune(k, v)
    for _ in range(3):
        wl.step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / a.steps
    ctx.profile_kernels(True)
    wl.step()
    fam = {}
    for f in getattr(wl, 'families', ('classify',)):
        try:
            fam[f] = round(ctx.last_kernel_ms(f) * 1e3, 1)
        except RuntimeError:
            pass
    ctx.profile_kernels(False)
    print(cfg, 'step %.1f us' % (dt * 1e6), fam, flush=True)
