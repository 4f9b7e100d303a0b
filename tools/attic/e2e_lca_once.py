#!/usr/bin/env python3
"""The config-3 command twice (for profilers): writes the inputs under /dev/shm
unless they are there already, prints the seconds of each call."""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from woltka_amd import synth, workflow  # noqa: E402

d = '/dev/shm/e2e_lca_once'
if not os.path.exists(d + '/nodes.dmp'):
    os.makedirs(d + '/in', exist_ok=True)
    rng = np.random.default_rng(1003)
    p = synth.as_sets(synth.lca_problem(rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=50_000_000,
                                        with_names=False))
    bench.write_sam_lca(d + '/in/S1.sam', p, 50_000_000)
    bench.write_nodes_dmp(d + '/nodes.dmp', p['hier'])
    del p
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow(d + '/in', d + '/out', input_fmt='sam', output_fmt=False, nodes_fps=[d + '/nodes.dmp'],
                          ranks='phylum,genus,species')
    print(f'e2e {time.perf_counter() - t0:.3f} s', flush=True)
