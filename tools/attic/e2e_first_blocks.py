import os, sys, time, io, contextlib, shutil
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from woltka_amd import synth, workflow
d = '/dev/shm/e2p4'
shutil.rmtree(d, ignore_errors=True); os.makedirs(d + '/in')
rng = np.random.default_rng(1003)
p = synth.as_sets(synth.lca_problem(rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=20_000_000, with_names=False))
n_rec, size = bench.write_sam_lca(d + '/in/S1.sam', p, 20_000_000)
bench.write_nodes_dmp(d + '/nodes.dmp', p['hier'])
os.environ['WOLTKA_DTOK_TIMING'] = '1'
for rep in range(2):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow(d + '/in', d + '/out', input_fmt='sam', output_fmt=False, nodes_fps=[d + '/nodes.dmp'], ranks='phylum,genus,species')
    print('run', rep, round(time.perf_counter() - t0, 3), flush=True)
shutil.rmtree(d, ignore_errors=True)
