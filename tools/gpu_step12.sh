#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
{
for rep in 1 2; do for b in 0 8; do for c in synthetic_1M_spheres synthetic_1M_triangles; do
  echo -n "bands=$b $c: "
  RTX_XCD_BANDS=$b timeout -k 10 300 python3 bench.py --config $c --steps 2 --no-cpu-baseline --no-pmc --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['image_check']['max_rel_diff'], d['image_check']['rays_equal'], d['image_check']['paths_equal'])"
done; done; done
} | tee gpurun_out/r03/xcd_bands_ab.txt
