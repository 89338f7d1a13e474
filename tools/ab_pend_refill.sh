#!/bin/bash
# A/B helper for the GPU box: the planar scenes (Cornell box, quads) for a grid of the two thresholds of the node walk
# (RTX_PEND_AT: waiting lanes before the surface tests, RTX_REFILL_AT: idle lanes before a refill).
for c in cornell_box_20k quads; do
for p in 28 32 36; do for r in 8 12 16; do
RTX_PEND_AT=$p RTX_REFILL_AT=$r timeout -k 10 300 python3 -u tools/config_sweep.py --only $c 2>/dev/null | LABEL="pend=$p refill=$r" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LABEL"], d["config"], d["mrays_per_s"], "Mrays/s", d["render_s"], "s", flush=True)'
done; done; done
