#!/bin/bash
# Copies what tools/round_profile.sh TAG left under gpurun_out/TAG (scratch, not tracked) into profiles/ (tracked), named per round.
#   tools/collect_profiles.sh TAG
set -e
TAG=${1:-r03}
OUT=gpurun_out/$TAG
for f in "$OUT"/bench_*.json; do cp "$f" "profiles/${TAG}_$(basename "$f")"; done
cp "$OUT/kernel_stats.csv" "profiles/${TAG}_kernel_stats.csv"
cp "$OUT/gputests.log" "profiles/${TAG}_gputests.log"
cp "$OUT/pmc/pmc_summary.json" "profiles/${TAG}_pmc_summary.json"
cp "$OUT/pmc/pmc_summary.md" "profiles/${TAG}_pmc_summary.md"
[ -f "$OUT/pmc/hbm_traffic.json" ] && cp "$OUT/pmc/hbm_traffic.json" profiles/hbm_traffic.json
ls -la profiles | grep "${TAG}_" | wc -l
