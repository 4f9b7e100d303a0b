import os, sys, time, io, contextlib, shutil
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from woltka_amd import synth, workflow, classify
d = '/dev/shm/e2b'
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + '/in')
rng = np.random.default_rng(1003)
p = synth.as_sets(synth.lca_problem(rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=50_000_000, with_names=False))
n_rec, size = bench.write_sam_lca(d + '/in/S1.sam', p, 50_000_000)
bench.write_nodes_dmp(d + '/nodes.dmp', p['hier'])
def run(tag):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow(d + '/in', d + '/out', input_fmt='sam', output_fmt=False, nodes_fps=[d + '/nodes.dmp'], ranks='phylum,genus,species')
    t = time.perf_counter() - t0
    print(f'{tag}: {t:.3f} s  {n_rec / t / 1e6:.1f} M records/s', flush=True)
for blk in (1 << 26, 1 << 27, 1 << 28, 1 << 25):
    classify.Engine.DTOK_BLOCK = blk
    for rep in range(3):
        run(f'block {blk >> 20} MB')
shutil.rmtree(d, ignore_errors=True)
