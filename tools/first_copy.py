#!/usr/bin/env python3
"""What the first copy of a block to the device costs in a fresh context
(the reader's first `wk_dtok_copy_ahead`), step by step on stderr."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

for rep in range(3):
    t0 = time.perf_counter()
    ctx = nat.Context(0)
    t1 = time.perf_counter()
    buf = ctx.host_alloc(65 << 20, np.uint8)
    buf[:1 << 20] = 10
    t2 = time.perf_counter()
    ctx.dtok_format('sam')
    out = {}

    def work():
        a = time.perf_counter()
        t = ctx.dtok_copy_ahead(buf, 0, 1 << 20)
        b = time.perf_counter()
        t = ctx.dtok_copy_ahead(buf, 0, 1 << 20)
        out['t'] = (b - a, time.perf_counter() - b)
    th = threading.Thread(target=work)
    th.start()
    th.join()
    print('context %.1f ms, pinned 65 MB %.1f ms, first copy call %.1f ms, second %.2f ms'
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, out['t'][0] * 1e3, out['t'][1] * 1e3), file=sys.stderr)
    ctx.tune('lap_print', 1)
    ctx.sync()
    ctx.dtok_copy_drop()
    del ctx
