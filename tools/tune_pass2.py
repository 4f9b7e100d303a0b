#!/usr/bin/env python3
"""Launch-geometry sweep of the generic (second) pass on a bench workload."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

wlname = sys.argv[1] if len(sys.argv) > 1 else 'lca'
ctx = nat.Context(0)
wl = bench.WORKLOADS[wlname](ctx, 1002, 1.0)
for threads, per_cu, slots in ((1024, 1, 8192), (1024, 1, 4096), (512, 2, 4096),
                               (512, 2, 2048), (512, 1, 8192), (256, 4, 2048),
                               (256, 4, 1024), (768, 1, 8192)):
    ctx.tune('threads', threads)
    ctx.tune('blocks_per_cu', per_cu)
    ctx.tune('lds_slots', slots)
    for _ in range(2):
        wl.step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        wl.step()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    print(f'threads {threads} x {per_cu}/CU, {slots} slots: {dt * 1e3:.2f} ms/step', flush=True)
