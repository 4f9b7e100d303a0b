#!/usr/bin/env python3
"""Kernel times of the coord-match step on the config-4 batch, with the log of
the gene tally by gene range (product) and hashed (range_log=0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
wl = bench.OrdinalWorkload(ctx, 1002, float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
for flag in (1, 0, 1):
    ctx.tune('range_log', flag)
    for _ in range(3):
        wl.step()
    wl.sync()
    k = bench.kernel_times(wl, n=3, burst=4)
    ctx.counts_clear()
    wl.step()
    keys, vals = ctx.counts_fetch()
    print('range_log', flag, k, 'keys', keys.size, 'sum', int(vals.sum()), 'xor', int(np.bitwise_xor.reduce(keys.astype(np.uint64) * np.uint64(31) + vals.astype(np.uint64))), flush=True)
    ctx.counts_clear()
