#!/bin/bash
# Copy what tools/prof_all.sh left under gpurun_out/<tag>_<workload>/ into profiles/
# (kernel stats and PMC summaries named per round, the traffic figures bench.py reads):
#   tools/collect_profiles.sh <tag> <round prefix, e.g. r04>
tag=$1; rnd=$2
for d in gpurun_out/${tag}_*; do
  [ -d "$d" ] || continue
  wl=${d#gpurun_out/${tag}_}
  [ -f "$d/kernel_stats.csv" ] && cp "$d/kernel_stats.csv" "profiles/${rnd}_${wl}_kernel_stats.csv"
  [ -s "$d/pmc_summary.txt" ] && cp "$d/pmc_summary.txt" "profiles/${rnd}_${wl}_pmc_summary.txt"
  [ -f "$d/traffic.json" ] && cp "$d/traffic.json" "profiles/traffic_${wl}.json"
  echo "$wl: $(ls $d | tr '\n' ' ')"
done
