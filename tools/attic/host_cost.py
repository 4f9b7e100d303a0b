#!/usr/bin/env python3
"""Host-side cost of one step (launch calls only) vs GPU time per step."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'flat'](ctx, 1002, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
for o in sys.argv[3:]:
    k, v = o.split('=')
    ctx.tune(k, int(v))
for _ in range(5):
    wl.step()
ctx.sync()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    wl.step()
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print('host issue %.1f us/step, wall %.1f us/step' % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
ctx.timer_begin()
for _ in range(N):
    wl.step()
ctx.timer_end()
print('gpu events %.1f us/step' % (ctx.timer_ms() / N * 1e3))
