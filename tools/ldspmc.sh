#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/ldspmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in lca ordinal; do
  timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/$wl -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu > $OUT/$wl.log 2>&1
  echo "$wl rc=$?"
  f=$(find $OUT/$wl -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pmc_$wl.csv
  rm -rf $OUT/$wl
done
cd $R
for wl in lca ordinal; do mkdir -p gpurun_out/ldspmc/tmp_$wl; cp gpurun_out/ldspmc/pmc_$wl.csv gpurun_out/ldspmc/tmp_$wl/pmc1.csv; python tools/pmc_summary.py gpurun_out/ldspmc/tmp_$wl > gpurun_out/ldspmc/summary_$wl.txt; rm -rf gpurun_out/ldspmc/tmp_$wl; done
rm -f gpurun_out/ldspmc/pmc_*.csv
grep -A9 "classify_kernel<true, true>\|classify_single" gpurun_out/ldspmc/summary_*.txt | head -80
