#!/usr/bin/env python3
"""Kernel-family times of a workload under the -DWK_ABLATE flags
(WOLTKA_HIP_LIB=woltka_amd/libwoltka_hip_ablate.so): 1 = drop counts,
2 = skip LDS flush, 8 = loads only, 16 = drop LDS-cache misses."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ctx = nat.Context(0)
wl = bench.WORKLOADS[sys.argv[1]](ctx, 1002, float(sys.argv[2]))
for o in sys.argv[3:]:
    k, v = o.split('=')
    ctx.tune(k, int(v))
fams = ('classify', 'leftover', 'partition_merge', 'dense_merge', 'match_count', 'match_write')
for abl in (0, 1, 16, 17, 8):
    ctx.tune('ablate', abl)
    for _ in range(2):
        wl.step()
    ctx.sync()
    ctx.profile_kernels(True)
    acc = {}
    for _ in range(3):
        wl.step()
        for f in fams:
            try:
                acc.setdefault(f, []).append(ctx.last_kernel_ms(f))
            except RuntimeError:
                pass
    ctx.profile_kernels(False)
    print('ablate', abl, {f: round(min(v), 3) for f, v in acc.items() if v}, flush=True)
