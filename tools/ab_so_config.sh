#!/bin/bash
# A/B helper for the GPU box: times one bench configuration with each of the prebuilt libraries given (abtmp/<name>.so), alternating,
# so that variants are compared on one machine.   tools/ab_so_config.sh <config> <spp or 0> <rounds> name1 name2 ...
set -e
cfg="$1"; spp="$2"; rounds="$3"; shift 3
for r in $(seq 1 "$rounds"); do
  for n in "$@"; do
    cp "abtmp/$n.so" raytracing_amd/librtx.so
    timeout -k 10 300 python3 -u bench.py --config "$cfg" --no-cpu-baseline --no-pmc --steps 2 --warmup 1 --spp "$spp" 2>/dev/null | LABEL="$n" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LABEL"], d["config"]["workload"][:40], d["ms_per_step"], "ms", d["value"], "Mrays/s", flush=True)'
  done
done
