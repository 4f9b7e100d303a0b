#!/bin/bash
# where the end-to-end time of the device-tokenizer route goes: kernel trace of
# one `woltka classify` run + block-size sweep
out=gpurun_out/${1:-r03p}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_dtok.py -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $out/pytest.log
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
cat $out/gen.log
cat > /tmp/run_e2e.py <<'PY'
import sys, time, io, contextlib
sys.path.insert(0, '.')
from woltka_amd import workflow
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        workflow.workflow('/dev/shm/e2e/in', '/dev/shm/e2e/out', input_fmt='sam', output_fmt=False,
                          nodes_fps=['/dev/shm/e2e/nodes.dmp'], ranks='phylum,genus,species')
    print(f'e2e {time.perf_counter() - t0:.3f} s', flush=True)
PY
for blk in 16777216 67108864 268435456; do
  echo "== DTOK block $blk"; WOLTKA_DTOK_BLOCK=$blk python /tmp/run_e2e.py 3
done
echo "== host tokenizer (mmap)"; WOLTKA_NO_DTOK=1 WOLTKA_READ=mmap python /tmp/run_e2e.py 2
echo "== host tokenizer (pread)"; WOLTKA_NO_DTOK=1 python /tmp/run_e2e.py 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o e2e -- python /tmp/run_e2e.py 1 > $GRAFT_REPO_ROOT/$out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/prof -name "*kernel_stats*" | head -3
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
rm -rf /dev/shm/e2e
