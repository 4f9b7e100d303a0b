#!/usr/bin/env python3
"""cProfile of the whole `classify --coords` call on the config-4 text (host
side costs around the streaming): writes the inputs once, runs the call twice
(warm), profiles the second."""
import contextlib
import cProfile
import io
import os
import pstats
import shutil
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import synth, workflow  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
d = '/dev/shm/e2e_ord'
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + '/in')
rng = np.random.default_rng(1002)
p = synth.ordinal_problem(rng, n_pairs=int(50_000_000 * frac))
sam, coords, n_rec, size = bench.write_ordinal_inputs(d + '/in', p, int(p['n_reads']))
os.replace(coords, d + '/coords.txt')


def run():
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow(d + '/in', d + '/out.tsv', input_fmt='sam', coords_fp=d + '/coords.txt')
    return time.perf_counter() - t0


print('warm', run(), flush=True)
pr = cProfile.Profile()
pr.enable()
t = run()
pr.disable()
print('profiled', t, n_rec / t / 1e6, 'M records/s')
pstats.Stats(pr).sort_stats('cumulative').print_stats(60)
shutil.rmtree(d, ignore_errors=True)
