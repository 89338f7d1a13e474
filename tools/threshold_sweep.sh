#!/bin/bash
# thresholds of the resident kernel, re-measured with the assembly node loop (100 spp per call)
for p in 12 16 20 24; do for b in 48 52 56; do
  RTX_PEND_AT=$p RTX_BOUNCE_AT=$b timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame --steps 3 --warmup 1 --spp 100 2>/dev/null | P=$p B=$b python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("pend", os.environ["P"], "bounce", os.environ["B"], d["ms_per_step"], "ms", flush=True)'
done; done
