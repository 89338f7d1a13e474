#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r03/gputests_a.log 2>&1 || { tail -30 gpurun_out/r03/gputests_a.log; exit 1; }
tail -3 gpurun_out/r03/gputests_a.log
for c in cornell_box quads simple_light hello_triangle synthetic_1M_triangles synthetic_1M_spheres cover_spheres; do
  for tb in 1 0; do
    RTX_TIGHT_BOXES=$tb timeout -k 10 300 python3 bench.py --config $c --steps 2 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c tight=$tb', d['value'], 'Mrays/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r03/cull_ab.txt
