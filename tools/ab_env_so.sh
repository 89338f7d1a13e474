#!/bin/bash
# A/B helper for the GPU box: one bench configuration under (library, environment) pairs.
#   tools/ab_env_so.sh <config> <spp or 0> "name|VAR=v VAR2=w" "name2|..." ...
cfg="$1"; spp="$2"; shift 2
for pair in "$@"; do
  n="${pair%%|*}"; e="${pair#*|}"
  cp "abtmp/$n.so" raytracing_amd/librtx.so
  env $e timeout -k 10 300 python3 -u bench.py --config "$cfg" --no-cpu-baseline --no-pmc --steps 2 --warmup 1 --spp "$spp" 2>/dev/null | LABEL="$n $e" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LABEL"], d["config"]["workload"][:30], d["ms_per_step"], "ms", d["value"], "Mrays/s", flush=True)'
done
