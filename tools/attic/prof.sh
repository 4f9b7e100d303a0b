#!/bin/bash
# rocprofv3 helper (GPU box): kernel trace + stats, then PMC passes, CSV output.
# usage: tools/prof.sh <outdir-under-gpurun_out> <python args...>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p "$OUT"
S=$R/$1; shift   # script path relative to the repo root
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python "$S" "$@" > "$OUT/kt.log" 2>&1
echo "kernel-trace rc=$?"
f=$(find "$OUT/kt" -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cut -c1-160 "$OUT/kernel_stats.csv" | head -12
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- python "$S" "$@" > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?"
  f=$(find "$OUT/pmc$i" -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/pmc$i.csv"
done
python "$R/tools/pmc_summary.py" "$OUT" | tee "$OUT/pmc_summary.txt"
# keep only the small summaries
rm -rf "$OUT"/kt "$OUT"/pmc[0-9]
