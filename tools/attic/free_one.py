#!/usr/bin/env python3
"""A few passes of the free-rank stream on the config-3 batch (for profilers)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
wl = bench.LcaWorkload(ctx, 1002, float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
free = bench.LcaFreeWorkload(ctx, 0, 1.0, share=wl)
for _ in range(5):
    free.step()
free.sync()
