#!/bin/bash
# kernel trace of one `woltka classify` run through the device tokenizer
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
os.makedirs('/dev/shm/e2e/in', exist_ok=True)
print(bench.write_sam_lca('/dev/shm/e2e/in/S1.sam', p, 50_000_000))
bench.write_nodes_dmp('/dev/shm/e2e/nodes.dmp', p['hier'])
PY
cat > /tmp/run_e2e.py <<PY
import sys, time, io, contextlib
sys.path.insert(0, '$R')
from woltka_amd import workflow
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow('/dev/shm/e2e/in', '/dev/shm/e2e/out', input_fmt='sam', output_fmt=False,
                          nodes_fps=['/dev/shm/e2e/nodes.dmp'], ranks='phylum,genus,species')
    print(f'e2e {time.perf_counter() - t0:.3f} s', flush=True)
PY
python /tmp/run_e2e.py 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o e2e -- python /tmp/run_e2e.py 1 > $R/$out/rocprof.log 2>&1
cd $R
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/e2e_kernel_stats.csv && head -16 "$f"
rm -rf $out/prof /dev/shm/e2e
