#!/usr/bin/env python3
"""Time the config-3 step under several option sets (one workload build).
usage: python tools/weigh_sweep.py [scale] name=v,name=v ..."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 and '=' not in sys.argv[1] else 1.0
sets = [a for a in sys.argv[1:] if '=' in a] or ['weigh=1']
ctx = nat.Context(0)
wl = bench.LcaWorkload(ctx, seed=1002, scale=scale)
ctx.sync()
base = None
for spec in sets:
    opts = dict(kv.split('=') for kv in spec.split(','))
    for k, v in opts.items():
        ctx.tune(k, int(v))
    for _ in range(3):
        wl.step()
    ctx.sync()
    ctx.timer_begin()
    for _ in range(10):
        wl.step()
    ctx.timer_end()
    ctx.sync()
    ms = ctx.timer_ms() / 10
    ctx.counts_clear()
    wl.step()
    try:
        keys, vals = ctx.counts_fetch()
    except ValueError:          # measurement knobs that break the counts
        print(f'{spec:50s} {ms:8.4f} ms/step  (counts invalid)', flush=True)
        ctx.counts_clear()
        continue
    k, v = nat.canonical_counts(keys, vals)
    sig = (int(k.size), int(v.sum() % (1 << 61)), int((k * v).sum() % (1 << 61)))
    if base is None:
        base = sig
    print(f'{spec:50s} {ms:8.4f} ms/step  table {"same" if sig == base else "DIFFERENT"}',
          flush=True)
