#!/usr/bin/env python3
"""Decompose the single-candidate pass with the -DWK_ABLATE build
(WOLTKA_HIP_LIB=woltka_amd/libwoltka_hip_ablate.so): 8 = loads only,
1 = no count, 2 = no flush."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'flat'](ctx, 1002, 1.0)
ctx.tune('split', 2)
for abl in (0, 1, 2, 3, 8, 10):
    ctx.tune('ablate', abl)
    for _ in range(3):
        wl.step()
    ctx.sync()
    ctx.profile_kernels(True)
    v = []
    for _ in range(10):
        wl.step()
        v.append(ctx.last_kernel_ms('classify') * 1e3)
    ctx.profile_kernels(False)
    print('ablate', abl, 'classify %.1f us (min %.1f)' % (sum(v) / len(v), min(v)), flush=True)
