#!/usr/bin/env python3
"""The headline's step (bench.TextLcaWorkload.step: every block of the sample's
text, resident in HBM, through the product's loop) timed at a small scale:
    WOLTKA_NO_LAG=1 / WOLTKA_LAG_POLL=1 python tools/lag_probe.py [scale]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
with nat.Context(0) as ctx:
    wl = bench.TextLcaWorkload(ctx, 1003, scale)
    for ab in [int(x, 0) for x in os.environ.get('FZ_ABLATE', '0').split(',')]:
      ctx.tune('fz_ablate', ab)
      for _ in range(3):
          wl.step()
      ctx.sync()
      best = None
      for rep in range(5):
          t0 = time.perf_counter()
          for _ in range(8):
              wl.step()
          ctx.sync()
          dt = (time.perf_counter() - t0) / 8
          best = dt if best is None else min(best, dt)
      print('%d blocks: %.3f ms per pass, %.1f us per block (NO_LAG=%s LAG_POLL=%s)' % (
          len(wl.blocks), best * 1e3, best * 1e6 / len(wl.blocks),
          os.environ.get('WOLTKA_NO_LAG'), os.environ.get('WOLTKA_LAG_POLL')), 'fz_ablate', ab)
    wl.close()
