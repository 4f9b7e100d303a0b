"""Does it matter on which NUMA node the process that feeds the GPU runs?
config 3 end to end, unpinned / pinned to the GPU's node / pinned to the other."""
import os, sys, time, io, contextlib, shutil
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from woltka_amd import synth, workflow, shard, _native as nat
mode = sys.argv[1]
if mode == 'near':
    print('pinned to', len(shard.pin_near_gpu(0) or []), 'cpus near', nat.device_pci_bus_id(0))
elif mode == 'far':
    near = set(shard.pin_near_gpu(0) or [])
    os.sched_setaffinity(0, set(range(os.cpu_count())) - near)
    print('pinned to', len(os.sched_getaffinity(0)), 'cpus on the other node')
d = '/dev/shm/e2n_' + mode
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + '/in')
rng = np.random.default_rng(1003)
p = synth.as_sets(synth.lca_problem(rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=20_000_000, with_names=False))
n_rec, size = bench.write_sam_lca(d + '/in/S1.sam', p, 20_000_000)
bench.write_nodes_dmp(d + '/nodes.dmp', p['hier'])
for rep in range(3):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow(d + '/in', d + '/out', input_fmt='sam', output_fmt=False, nodes_fps=[d + '/nodes.dmp'], ranks='phylum,genus,species')
    t = time.perf_counter() - t0
    print(f'{mode}: {t:.3f} s  {n_rec / t / 1e6:.1f} M records/s  {size / t / 1e9:.1f} GB/s', flush=True)
shutil.rmtree(d, ignore_errors=True)
