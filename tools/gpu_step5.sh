#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
for c in synthetic_1M_triangles cornell_box synthetic_1M_spheres; do
  timeout -k 10 600 python3 bench.py --config $c --steps 2 --no-cpu-baseline --no-cold-frame > gpurun_out/r03/try_$c.json 2> gpurun_out/r03/try_$c.err || { tail -20 gpurun_out/r03/try_$c.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open('gpurun_out/r03/try_$c.json')); r = d['roofline']
print('$c', d['value'], 'Mrays/s', d['ms_per_step'], 'ms;', r['bound'], r['frac'], 'Nv', r['nodes_per_ray'], 'Ns', r['sphere_tests_per_ray'], 'Np', r['planar_tests_per_ray'], 'ref', r['reference_walk']['nodes_per_ray'], r['reference_walk']['planar_tests_per_ray'])
print('  hbm', r['hbm']); print('  valu_issue', r['valu_issue']); print('  image_check', d.get('image_check')); print('  undecided', r['box_tests_left_to_f64_tiers'])
PY
done
