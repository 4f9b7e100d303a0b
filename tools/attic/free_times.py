#!/usr/bin/env python3
"""Kernel times of the per-read stream (csrc/wk_free.hpp) on the config-3 batch:
--rank free and rank genus under --above / --major 80 / --uniq."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
wl = bench.LcaWorkload(ctx, 1002, float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
free = bench.LcaFreeWorkload(ctx, 0, 1.0, share=wl)
print('free', {k: round(v, 4) for k, v in bench.kernel_times(free).items()}, flush=True)
for option in ('above', 'major', 'uniq'):
    w = bench.LcaOptionWorkload(ctx, option, wl)
    print(option, {k: round(v, 4) for k, v in bench.kernel_times(w).items()}, flush=True)
