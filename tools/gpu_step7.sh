#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
{
for b in 16 24 32 36; do for p in 8 12 16; do
  echo -n "resident bounce_at=$b pend_at=$p: "
  RTX_RESIDENT=1 RTX_BOUNCE_AT=$b RTX_PEND_AT=$p timeout -k 10 300 python3 bench.py --config cornell_box --spp 2000 --steps 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
for p in 16 24 32 40 48; do for r in 8 16 24; do
  echo -n "pooled pend_at=$p refill_at=$r: "
  RTX_PEND_AT=$p RTX_REFILL_AT=$r timeout -k 10 300 python3 bench.py --config cornell_box --spp 2000 --steps 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
} | tee gpurun_out/r03/cornell_sweep2.txt
