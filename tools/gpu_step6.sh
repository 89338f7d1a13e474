#!/bin/bash
# (1) what the 60 MiB scratch array of rtx_trace_closest_device adds to a cold process: A/B of the library with a 1024-ray array
# (2) the Cornell box in the resident form, thresholds swept, against the pooled form
set -e -o pipefail
mkdir -p gpurun_out/r03
{
for rep in 1 2 3; do
  for n in base smallscratch; do
    cp abtmp/$n.so raytracing_amd/librtx.so
    RTX_LOG_COLD=1 timeout -k 10 120 bin/rtx_example cover_spheres --width 1920 --spp 500 -o /tmp/cold.ppm 2>&1 | grep -E "rtx cold: hipGet|rtx cold: scene|Total runtime" | tr '\n' ';' | sed "s/^/$n: /"; echo
  done
done
} | tee gpurun_out/r03/scratch_cold_ab.txt
cp abtmp/base.so raytracing_amd/librtx.so
{
echo "pooled (default):"; timeout -k 10 300 python3 bench.py --config cornell_box --spp 2000 --steps 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
for b in 40 48 52 56 60; do for p in 16 24 32 40; do
  echo -n "resident bounce_at=$b pend_at=$p: "
  RTX_RESIDENT=1 RTX_BOUNCE_AT=$b RTX_PEND_AT=$p timeout -k 10 300 python3 bench.py --config cornell_box --spp 2000 --steps 2 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
} | tee gpurun_out/r03/cornell_resident_sweep.txt
