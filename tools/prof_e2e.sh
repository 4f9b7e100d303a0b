#!/bin/bash
# Profile one whole `woltka classify` call (workflow.workflow) on the GPU box:
# rocprofv3 kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate PMC
# passes, each around `python tools/e2e_once.py <kind> --run`; the inputs are
# generated first, outside the profiler.  Writes
#   gpurun_out/<tag>/kernel_stats.csv, pmc_summary.txt, e2e_profile.json
# usage: tools/prof_e2e.sh <tag> <kind> [reads]      (PMC=0: kernel stats only)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; KIND=$2; READS=${3:-0}
OUT=$R/gpurun_out/$TAG
D=${E2E_DIR:-/dev/shm/wk_e2e}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$R/tools/e2e_once.py" "$KIND" --dir "$D" --prepare --reads "$READS" > "$OUT/prepare.log" 2>&1 || { tail -5 "$OUT/prepare.log"; exit 1; }
tail -1 "$OUT/prepare.log"
# (one warm call outside the profiler: what the traced call costs without it)
python "$R/tools/e2e_once.py" "$KIND" --dir "$D" --run --reps 3 > "$OUT/plain.log" 2>&1
tail -1 "$OUT/plain.log"
CMD="python $R/tools/e2e_once.py $KIND --dir $D --run --reps ${REPS:-2}"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $CMD > "$OUT/kt.log" 2>&1
echo "kernel-trace rc=$?"
f=$(find "$OUT/kt" -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cut -c1-160 "$OUT/kernel_stats.csv" | head -14
if [ "${PMC:-1}" = 1 ]; then
  i=0
  for set in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 1500 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1
    echo "pmc$i rc=$?"
    f=$(find "$OUT/pmc$i" -name "*counter_collection.csv" 2>/dev/null | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/pmc$i.csv"
  done
  python "$R/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_summary.txt"
fi
cd "$R" && python tools/e2e_profile_summary.py "$OUT" "$KIND" "${REPS:-2}"
rm -rf "$OUT"/kt "$OUT"/pmc[0-9] "$OUT"/pmc[0-9].csv
[ "${KEEP_INPUTS:-0}" = 1 ] || rm -rf "$D"
