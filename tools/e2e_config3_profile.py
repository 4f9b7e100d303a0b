"""Where does config 3's end-to-end time go outside the block loop?  One warm
run under cProfile (top of the cumulative list) and one with the tokenizer's
laps (WOLTKA_DTOK_TIMING)."""
import os, sys, time, io, contextlib, shutil, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from woltka_amd import synth, workflow
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
d = '/dev/shm/e2p3'
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + '/in')
rng = np.random.default_rng(1003)
p = synth.as_sets(synth.lca_problem(rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=n_reads, with_names=False))
n_rec, size = bench.write_sam_lca(d + '/in/S1.sam', p, n_reads)
bench.write_nodes_dmp(d + '/nodes.dmp', p['hier'])
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow(d + '/in', d + '/out', input_fmt='sam', output_fmt=False, nodes_fps=[d + '/nodes.dmp'], ranks='phylum,genus,species')
for rep in range(2):
    t0 = time.perf_counter(); run(); t = time.perf_counter() - t0
    print(f'run {rep}: {t:.3f} s  {n_rec / t / 1e6:.1f} M records/s', flush=True)
os.environ['WOLTKA_DTOK_TIMING'] = '1'
t0 = time.perf_counter(); run(); t = time.perf_counter() - t0
print(f'timed run: {t:.3f} s', flush=True)
del os.environ['WOLTKA_DTOK_TIMING']
pr = cProfile.Profile()
pr.enable(); run(); pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue())
shutil.rmtree(d, ignore_errors=True)
