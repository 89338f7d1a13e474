#!/bin/bash
# A/B helper for the GPU box: times the headline scene with each of the prebuilt libraries given (abtmp/<name>.so), alternating,
# so that variants are compared on one machine.   tools/ab_so.sh <spp> <rounds> name1 name2 ...
set -e
spp="$1"; rounds="$2"; shift 2
for r in $(seq 1 "$rounds"); do
  for n in "$@"; do
    cp "abtmp/$n.so" raytracing_amd/librtx.so
    timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-pmc --steps 3 --warmup 1 --spp "$spp" 2>/dev/null | LABEL="$n" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LABEL"], d["ms_per_step"], "ms", d["value"], "Mrays/s", flush=True)'
  done
done
