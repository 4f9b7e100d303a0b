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
shapes = [  # (threads, cache slots, workgroups per CU)
    (1024, 8192, 1), (1024, 4096, 2), (1024, 2048, 2), (512, 4096, 3), (512, 2048, 4), (256, 1024, 6)]
for threads, slots, per_cu in shapes:
    ctx.tune('free_per_cu', per_cu)
    ctx.tune('free_threads', threads)
    ctx.tune('free_slots', slots)
    for _ in range(3):
        free.step()
    free.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        free.step()
    free.sync()
    dt = (time.perf_counter() - t0) / 20
    k = bench.kernel_times(free, n=2, burst=4)
    print(f'threads {threads} slots {slots} per_cu {per_cu}: pass {dt * 1e3:.3f} ms  kernels {k}', flush=True)
