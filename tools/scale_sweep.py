#!/usr/bin/env python3
"""Kernel time vs chunk size (slope = streaming rate, intercept = fixed cost)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

wlname = sys.argv[1] if len(sys.argv) > 1 else 'flat'
opts = [o.split('=') for o in sys.argv[2:]]
for scale in (0.1, 0.25, 0.5, 1.0, 2.0, 4.0):
    ctx = nat.Context(0)
    wl = bench.WORKLOADS[wlname](ctx, 1002, scale)
    for k, v in opts:
        ctx.tune(k, int(v))
    for _ in range(3):
        wl.step()
    ctx.sync()
    ctx.profile_kernels(True)
    fams = getattr(wl, 'families', ('classify',))
    acc = {f: [] for f in fams}
    for _ in range(10):
        wl.step()
        for f in fams:
            try:
                acc[f].append(ctx.last_kernel_ms(f) * 1e3)
            except RuntimeError:
                pass
    print('scale', scale, 'reads', wl.reads, {f: round(min(v), 1) for f, v in acc.items() if v}, flush=True)
    del wl, ctx
