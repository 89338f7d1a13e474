#!/bin/bash
# Runs on the GPU box: GPU tests, the headline bench, a rocprofv3 kernel-trace
# of the same bench command and the PMC passes.  Usage: tools/round_profile.sh TAG
# Every GPU step has its own time limit and the script stops at the first failure.
set -e -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== gpu tests"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/gputests.log" 2>&1
tail -2 "$OUT/gputests.log"
echo "== bench"
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
echo "== rocprofv3 kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
cat "$OUT/bench_under_rocprof.json"
find "$OUT/stats" -name '*kernel_trace.csv' -delete
find "$OUT/stats" -name '*kernel_stats.csv' -exec head -12 {} \;
echo "== pmc"
timeout -k 10 1000 python3 tools/pmc_profile.py "$OUT/pmc" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc.log" 2>&1
tail -30 "$OUT/pmc.log"
find "$OUT/pmc" -name '*counter_collection.csv' -delete
