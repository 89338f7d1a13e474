#!/bin/bash
cp abtmp/walk2.so raytracing_amd/librtx.so
for c in synthetic_1M_spheres synthetic_1M_triangles; do
for e in "RTX_NODE16=1 RTX_BATCH_F64_BOXES=0" "RTX_NODE16=1 RTX_BATCH_F64_BOXES=1" "RTX_NODE16=0 RTX_BATCH_F64_BOXES=0" "RTX_NODE16=0 RTX_BATCH_F64_BOXES=1"; do
env $e timeout -k 10 300 python3 -u bench.py --config $c --no-cpu-baseline --no-pmc --steps 2 --warmup 1 2>/dev/null | LABEL="$e" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LABEL"], d["config"]["workload"][:30], d["ms_per_step"], "ms", d["value"], "Mrays/s", flush=True)'
done; done
