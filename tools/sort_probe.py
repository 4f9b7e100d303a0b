#!/usr/bin/env python3
"""The sort of the coord-match (csrc/wk_stripe.hpp) by itself, for rocprofv3:
the bench's configs[3] chunk (or its first `--reads`) staged and counted
`--reps` times, the sort redone every time."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from woltka_amd import synth  # noqa: E402
from woltka_amd import _native as nat  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reads', type=int, default=0)
ap.add_argument('--reps', type=int, default=3)
a = ap.parse_args()
ctx = nat.Context(0)
rng = np.random.default_rng(1002)
p = synth.ordinal_problem(rng, n_pairs=50_000_000)
ctx.set_genes(p['genome_off'], p['gstart'], p['gend'], p['gene_feature'])
ctx.counts_reserve(1 << 22)
n = a.reads or int(p['n_reads'])
hoff = p['hoff'][:n + 1]
h = int(hoff[-1])
jobs = [nat.Job(nat.MODE_NONE, 0, 0, 0, 0.0)]
ctx.tune('stripes_min', 0)
ctx.profile_kernels(True)
for _ in range(a.reps):
    ctx.ordinal_stage(p['genome'][:h], p['beg'][:h], p['end'][:h],
                      p['length'][:h], hoff, 0.8)
    ctx.set_uniform_group(0)
    ctx.ordinal_count(jobs)
    print('reads', n, 'hits', h, 'sort_ms', round(ctx.last_kernel_ms('stripe_sort'), 3),
          'match_ms', round(ctx.last_kernel_ms('stripe_match'), 3))
