#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
{
for spp in 1 8 62 63 125 250 500; do
  echo -n "cover_spheres spp=$spp: "
  timeout -k 10 300 python3 bench.py --spp $spp --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step;', d['roofline']['avg_launch_ms'], 'ms launch;', d['value'], 'Mrays/s', d['config']['kernel'])"
done
} | tee gpurun_out/r03/spp_share_probe.txt
for c in synthetic_1M_triangles synthetic_1M_spheres; do
  timeout -k 10 500 python3 bench.py --config $c --spp 1000 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame > gpurun_out/r03/nominal_$c.json 2>/dev/null
  python3 -c "import json; d=json.load(open('gpurun_out/r03/nominal_$c.json')); print('$c nominal', d['config']['nominal'], d['ms_per_step'], 'ms', d['value'], 'Mrays/s')"
done
RTX_LOG_COLD=1 timeout -k 10 120 bin/rtx_example cover_spheres --width 1920 --spp 500 -o /tmp/cold.ppm 2>&1 | tail -8
RTX_LOG_COLD=1 timeout -k 10 120 bin/rtx_example cover_spheres --width 1920 --spp 500 -o /tmp/cold.ppm 2>&1 | tail -8
