#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
T0=$(date +%s)
timeout -k 10 500 python3 bench.py > gpurun_out/r03/bench_try.json 2> gpurun_out/r03/bench_try.err || { tail -30 gpurun_out/r03/bench_try.err; exit 1; }
echo "bench wall: $(( $(date +%s) - T0 )) s"
cat gpurun_out/r03/bench_try.json
