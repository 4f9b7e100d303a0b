#!/usr/bin/env python3
"""Large-chunk self-consistency: one config-3-shaped chunk at --scale times
SURVEY §8d's size classified with the two-class split and with the generic
kernel alone must give the same count table (sizes where the 32-bit offsets
of the first pass are near their limits, and beyond them)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
ctx = nat.Context(0)
t0 = time.time()
wl = bench.WORKLOADS['lca'](ctx, 1003, scale)
print(f'{wl.records} records, {wl.reads} reads staged in {time.time() - t0:.0f} s', flush=True)
tables = []
for split in (1, 0):
    ctx.tune('split', split)
    ctx.counts_clear()
    ctx.reset_stats()
    t0 = time.time()
    wl.step()
    ctx.sync()
    dt = time.time() - t0
    keys, vals = nat.canonical_counts(*ctx.counts_fetch())
    st = ctx.stats()
    print(f'split={split}: {dt * 1e3:.1f} ms, {keys.size} keys, stats {st["n_reads"]} reads '
          f'{st["n_records"]} records', flush=True)
    tables.append((keys, vals, st['n_reads'], st['n_records']))
same = (np.array_equal(tables[0][0], tables[1][0]) and np.array_equal(tables[0][1], tables[1][1])
        and tables[0][2:] == tables[1][2:])
print('IDENTICAL' if same else 'DIFFERENT')
sys.exit(0 if same else 1)
