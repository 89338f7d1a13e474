#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
{
for p in 8 16 24 32; do for r in 16 32 48; do
  echo -n "triangles pend_at=$p refill_at=$r: "
  RTX_PEND_AT=$p RTX_REFILL_AT=$r timeout -k 10 300 python3 bench.py --config synthetic_1M_triangles --steps 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
for q in 0 1; do
  echo -n "triangles node16=$q: "
  RTX_NODE16=$q timeout -k 10 300 python3 bench.py --config synthetic_1M_triangles --steps 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
for p in 16 24 32; do for r in 16 24 32; do
  echo -n "quads pend_at=$p refill_at=$r: "
  RTX_PEND_AT=$p RTX_REFILL_AT=$r timeout -k 10 300 python3 bench.py --config quads --spp 5000 --steps 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['kernel'])"
done; done
} | tee gpurun_out/r03/planar_threshold_sweep.txt
