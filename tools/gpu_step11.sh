#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_large_and_uv.py tests/test_gpu_cull_boxes.py -m gpu -x -q --timeout 400 --timeout-method thread 2>&1 | tail -4
{
for rep in 1 2; do for n in base pair; do for c in synthetic_1M_spheres synthetic_1M_triangles; do
  cp abtmp/$n.so raytracing_amd/librtx.so
  echo -n "$n $c: "
  timeout -k 10 300 python3 bench.py --config $c --steps 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done; done
} | tee gpurun_out/r03/pair_ab.txt
cp abtmp/pair.so raytracing_amd/librtx.so
