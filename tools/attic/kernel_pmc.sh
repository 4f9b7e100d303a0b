#!/bin/bash
# PMC counters of one kernel family under a python command: two passes of SQ
# counters (rocprofv3 --pmc with --kernel-trace only), averaged per kernel name.
#   tools/kernel_pmc.sh <tag> <kernel substring> <python args...>
tag=$1; shift
pat=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
p=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
  p=$((p + 1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pmc$p" -o pmc -- python "$@" > "$out/pmc$p.log" 2>&1
done
python - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            a = acc[(r['Kernel_Name'][:60], r['Counter_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f'{k:60s} {c:24s} {v / n:16.1f}  (n={n})')
PY
