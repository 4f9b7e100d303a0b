#!/bin/bash
# Same-box A/B of the reader that trims SAM lines on their way into pinned memory
# (routes/device_text._trim_blocks) against the one that copies them whole:
# whole `woltka classify` calls on config 3's text with and without SEQ / QUAL.
#   bash tools/ab_trim.sh [out_dir]
out=${1:-gpurun_out/ab_trim}
mkdir -p "$out"
for kind in lca_seqqual lca; do
  d=/dev/shm/wk_e2e_$kind
  python tools/e2e_once.py $kind --dir $d --prepare > "$out/$kind.prepare.txt" 2>&1
  for mode in trim whole mapped trim; do
    case $mode in
      trim) env="";;
      whole) env="WOLTKA_NO_TRIM=1";;
      mapped) env="WOLTKA_TRIM_MAPPED=1";;
    esac
    echo "== $kind $mode" >> "$out/$kind.txt"
    env $env WOLTKA_DTOK_TIMING=1 python tools/e2e_once.py $kind --dir $d --run --reps 3 >> "$out/$kind.txt" 2>&1
  done
  rm -rf $d
done
