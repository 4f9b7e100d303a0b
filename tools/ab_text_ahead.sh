#!/bin/bash
# The reader started before the hierarchy (routes.device_text.start_text_ahead)
# against the same command without it, alternating, per kind:
#   tools/ab_text_ahead.sh <out file> <kind> [<kind> ...]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; shift
D=${E2E_DIR:-/dev/shm/wk_e2e}
: > "$OUT"
for k in "$@"; do
  python "$R/tools/e2e_once.py" "$k" --dir "$D" --prepare --reads "${READS:-0}" >> "$OUT" 2>&1
  for round in 1 2; do
    for off in "" 1; do
      echo "== $k WOLTKA_NO_TEXT_AHEAD='$off'" >> "$OUT"
      WOLTKA_NO_TEXT_AHEAD=$off python "$R/tools/e2e_once.py" "$k" --dir "$D" --run --reps "${REPS:-3}" >> "$OUT" 2>&1
    done
  done
  echo "== $k timing" >> "$OUT"
  WOLTKA_DTOK_TIMING=1 python "$R/tools/e2e_once.py" "$k" --dir "$D" --run --reps 2 2>&1 | grep -v "^\[wk_hier\]" | cut -c1-700 >> "$OUT"
  rm -rf "$D/$k"
done
rm -rf "$D"
grep -h '^== \|^{"kind"\|Error\|error\|^\[dtok\] [0-9]\|^\[wk\]' "$OUT" | cut -c1-400
