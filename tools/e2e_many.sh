#!/bin/bash
# Several end-to-end legs in a row (tools/e2e_once.py), timings only:
#   tools/e2e_many.sh <out file> <kind> [<kind> ...]      READS=<n> to shrink
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; shift
D=${E2E_DIR:-/dev/shm/wk_e2e}
: > "$OUT"
for k in "$@"; do
  python "$R/tools/e2e_once.py" "$k" --dir "$D" --prepare --reads "${READS:-0}" >> "$OUT" 2>&1
  python "$R/tools/e2e_once.py" "$k" --dir "$D" --run --reps "${REPS:-3}" >> "$OUT" 2>&1
  rm -rf "$D/$k"
done
rm -rf "$D"
grep -h '^{"kind"\|Error\|error' "$OUT" | cut -c1-300
