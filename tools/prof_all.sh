#!/bin/bash
# rocprofv3 kernel stats + PMC passes (tools/prof_bench.sh) of every bench
# workload; results under gpurun_out/<tag>_<workload>/
#   tools/prof_all.sh <tag> [workloads...]
tag=${1:-r03prof}; shift
wls=${*:-"lca_text lca lca_free lca_above lca_major lca_uniq ordinal flat"}
for wl in $wls; do
  case $wl in
    lca_text) kern=dtok_fused_kernel ;;
    lca) kern=weigh_streams ;;
    lca_free|lca_above|lca_major|lca_uniq|lca_above3) kern=free_stream ;;
    ordinal) kern=stripe_match ;;
    flat) kern=count_subjects ;;
  esac
  # (FULL="lca ordinal": every PMC set for these, the two traffic passes only for the others)
  sets=all
  if [ -n "${FULL:-}" ]; then case " $FULL " in *" $wl "*) sets=all ;; *) sets=traffic ;; esac; fi
  echo "== $wl ($kern, PMC sets: $sets)"
  PMC_SETS=$sets bash tools/prof_bench.sh ${tag}_$wl $wl 1.0 $kern 2>&1 | tail -12
done
