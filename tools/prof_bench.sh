#!/bin/bash
# Profile the bench command on the GPU box: rocprofv3 kernel trace + stats, then
# FETCH_SIZE and WRITE_SIZE in separate PMC passes; writes
#   gpurun_out/<tag>/kernel_stats.csv, pmc_summary.txt, traffic.json
# usage: tools/prof_bench.sh <tag> <workload> <scale> <kernel-name-substring>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; WL=$2; SCALE=$3; KERN=$4
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload $WL --scale $SCALE --steps ${STEPS:-20} --warmup 3 --headline-only --passes ${PASSES:-2} ${EXTRA:-}"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $CMD > "$OUT/kt.log" 2>&1
echo "kernel-trace rc=$?"
f=$(find "$OUT/kt" -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && cut -c1-150 "$OUT/kernel_stats.csv" | head -8
i=0
# (PMC_SETS=traffic: the two passes the traffic figure needs, nothing else)
sets=("FETCH_SIZE" "WRITE_SIZE")
[ "${PMC_SETS:-all}" = all ] && sets+=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU")
for set in "${sets[@]}"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?"
  f=$(find "$OUT/pmc$i" -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/pmc$i.csv"
done
python "$R/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_summary.txt"
cd "$R" && python - "$OUT" "$WL" "$SCALE" "$KERN" <<'PY'
import csv, glob, json, os, sys
out, wl, scale, kern = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
vals = {'FETCH_SIZE': [], 'WRITE_SIZE': []}
for fp in glob.glob(os.path.join(out, 'pmc*.csv')):
    for row in csv.DictReader(open(fp)):
        if kern in row.get('Kernel_Name', '') and row['Counter_Name'] in vals:
            vals[row['Counter_Name']].append(float(row['Counter_Value']))
if vals['FETCH_SIZE'] and vals['WRITE_SIZE']:
    fetch_kb = sum(vals['FETCH_SIZE']) / len(vals['FETCH_SIZE'])
    write_kb = sum(vals['WRITE_SIZE']) / len(vals['WRITE_SIZE'])
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
    # reports half of the bytes of a coalesced streaming read -> doubled
    sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
    from woltka_amd import _native as nat
    import __graft_entry__ as ge
    t = {'workload': wl, 'scale': scale, 'kernel': kern, 'launches': len(vals['FETCH_SIZE']), 'build_id': nat.build_id(),
         'device_digest': ge.device_digest(),
         'FETCH_SIZE_KiB_mean': fetch_kb, 'WRITE_SIZE_KiB_mean': write_kb,
         'hbm_bytes_per_launch': int(2 * fetch_kb * 1024 + write_kb * 1024),
         'note': 'FETCH_SIZE doubled (gfx950 correction); separate PMC passes'}
    json.dump(t, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
    print(json.dumps(t))
PY
rm -rf "$OUT"/kt "$OUT"/pmc[0-9] "$OUT"/pmc[0-9].csv
