#!/bin/bash
# dtok_parse_kernel<false> on config 3's text and on config 5's (80 us against 330 us per block):
# the same counters around one `woltka classify` call each, the one-kernel tokenizer switched off for config 3.
#   bash tools/pmc_parse_two_inputs.sh <out dir>
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/${1:-gpurun_out/pmc_parse}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kind in lca twopass1; do
  D=/dev/shm/wk_pmc_$kind
  python "$R/tools/e2e_once.py" $kind --dir $D --prepare ${READS:+--reads $READS} > "$OUT/$kind.prepare.log" 2>&1
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    WOLTKA_NO_FUSED=1 timeout 900 rocprofv3 --pmc $set --output-format csv -d "$OUT/${kind}_pmc$i" -- python "$R/tools/e2e_once.py" $kind --dir $D --run --reps 1 > "$OUT/${kind}_pmc$i.log" 2>&1
    echo "$kind pmc$i rc=$?"
    f=$(find "$OUT/${kind}_pmc$i" -name "*counter_collection.csv" 2>/dev/null | head -1)
    [ -n "$f" ] && python - "$f" "$kind" <<'PY' >> "$OUT/summary.txt"
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if 'dtok_parse_kernel<false>' in row.get('Kernel_Name', ''):
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f'{sys.argv[2]:9s} dtok_parse<false> {k:32s} n={len(v):5d} mean={sum(v) / len(v):16.1f}')
PY
    rm -rf "$OUT/${kind}_pmc$i"
  done
  rm -rf $D
done
cat "$OUT/summary.txt"
