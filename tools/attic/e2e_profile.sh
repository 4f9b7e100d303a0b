#!/bin/bash
# where the host side of the device-tokenizer route spends its time
out=gpurun_out/${1:-r03p}
mkdir -p $out
R=$GRAFT_REPO_ROOT
python - <<'PY' > $out/gen.log 2>&1
import sys, os, numpy as np
sys.path.insert(0, '.')
import bench
from woltka_amd import synth
rng = np.random.default_rng(1003)
p = synth.as_sets(synth.lca_problem(rng, n_nodes=2_000_000, n_subjects=100_000, n_reads=50_000_000, with_names=False))
for d in ('/dev/shm/e2e', '/tmp/e2e'):
    os.makedirs(d + '/in', exist_ok=True)
    print(bench.write_sam_lca(d + '/in/S1.sam', p, 50_000_000))
    bench.write_nodes_dmp(d + '/nodes.dmp', p['hier'])
PY
cat > /tmp/run_e2e.py <<PY
import sys, time, io, contextlib
sys.path.insert(0, '$R')
from woltka_amd import workflow
d = sys.argv[2]
for rep in range(int(sys.argv[1])):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow(d + '/in', d + '/out', input_fmt='sam', output_fmt=False,
                          nodes_fps=[d + '/nodes.dmp'], ranks='phylum,genus,species')
    print(f'e2e {time.perf_counter() - t0:.3f} s', flush=True)
PY
export WOLTKA_DTOK_TIMING=1
for d in /dev/shm/e2e /tmp/e2e; do
  echo "== dir $d"; python /tmp/run_e2e.py 3 $d
  for t in 8 32 64; do echo "== dir $d, tokenizer_threads $t (reader = half)"; WOLTKA_TOK_THREADS=$t python /tmp/run_e2e.py 2 $d; done
done
rm -rf /dev/shm/e2e /tmp/e2e
