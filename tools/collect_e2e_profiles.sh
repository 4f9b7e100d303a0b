#!/bin/bash
# Copy what tools/prof_e2e.sh left under gpurun_out/<tag>_<kind>/ into profiles/:
#   tools/collect_e2e_profiles.sh <tag> <round prefix, e.g. r05>
tag=$1; rnd=$2
for d in gpurun_out/${tag}_*; do
  [ -d "$d" ] || continue
  kind=${d#gpurun_out/${tag}_}
  [ -f "$d/kernel_stats.csv" ] && cp "$d/kernel_stats.csv" "profiles/${rnd}_e2e_${kind}_kernel_stats.csv"
  [ -s "$d/pmc_summary.txt" ] && cp "$d/pmc_summary.txt" "profiles/${rnd}_e2e_${kind}_pmc_summary.txt"
  [ -f "$d/e2e_profile.json" ] && cp "$d/e2e_profile.json" "profiles/${rnd}_e2e_${kind}_profile.json"
  echo "$kind: $(ls $d | tr '\n' ' ')"
done
