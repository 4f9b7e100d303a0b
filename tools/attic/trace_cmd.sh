R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace/kt -- python $R/tools/one_config.py --workload flat --steps 20 > $R/gpurun_out/trace/log.txt 2>&1
f=$(find $R/gpurun_out/trace/kt -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/tools/trace_gaps.py $f 14
rm -rf $R/gpurun_out/trace/kt
