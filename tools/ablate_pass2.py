#!/usr/bin/env python3
"""Second-pass prologue of config 2 under the -DWK_ABLATE flags: 32 = no merge,
64 = no compaction, 128 = merge without table adds."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
wl = bench.WORKLOADS['flat'](ctx, 1002, 1.0)
for abl in (0, 32, 64, 96, 128, 192):
    ctx.tune('ablate', abl)
    for _ in range(3):
        wl.step()
    ctx.sync()
    ctx.profile_kernels(True)
    v = []
    for _ in range(10):
        wl.step()
        v.append(ctx.last_kernel_ms('leftover') * 1e3)
    ctx.profile_kernels(False)
    print('ablate', abl, 'pass 2 %.1f us (min %.1f)' % (sum(v) / len(v), min(v)), flush=True)
