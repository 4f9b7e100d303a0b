#!/usr/bin/env python3
"""Run N steps of one bench workload with fixed knobs (for rocprofv3)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='flat')
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--opt', action='append', default=[], help='name=value')
a = ap.parse_args()
ctx = nat.Context(0)
wl = bench.WORKLOADS[a.workload](ctx, 1002, a.scale)
for o in a.opt:
    k, v = o.split('=')
    ctx.tune(k, int(v))
for _ in range(a.steps):
    wl.step()
ctx.sync()
ctx.profile_kernels(True)
wl.step()
for f in getattr(wl, 'families', ('classify',)):
    try:
        print(f, round(ctx.last_kernel_ms(f) * 1e3, 1), 'us')
    except RuntimeError:
        pass
print('done', wl.check())
