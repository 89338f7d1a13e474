#!/bin/bash
# Headline frame at 100 spp for a grid of the two thresholds of the node walk (idle lanes before a refill, waiting lanes
# before the surface tests).  Run on the GPU box from the repository root.
for r in 8 16 24 32; do for p in 8 16 24; do
  v=$(RTX_REFILL_AT=$r RTX_PEND_AT=$p timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --spp 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "refill_at $r pend_at $p ms $v"
done; done
