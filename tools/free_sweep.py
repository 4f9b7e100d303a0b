#!/usr/bin/env python3
"""Launch shapes of the free-rank stream (csrc/wk_free.hpp) on the config-3
batch: workgroups per CU x windows in flight per wave."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
wl = bench.LcaWorkload(ctx, 1002, float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
free = bench.LcaFreeWorkload(ctx, 0, 1.0, share=wl)
for per_cu in (1, 2, 3, 4, 6):
    for win in (2, 4, 8):
        ctx.set_option('free_per_cu', per_cu)
        ctx.set_option('free_windows', win)
        for _ in range(3):
            free.step()
        free.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            free.step()
        free.sync()
        dt = (time.perf_counter() - t0) / 20
        k = bench.kernel_times(free, n=2, burst=4)
        print(f'per_cu {per_cu} windows {win}: pass {dt * 1e3:.3f} ms  kernels {k}', flush=True)
