#!/bin/bash
# Kernel A/B aid for the GPU box: rebuilds librtx.so with -DRTX_REPLAY plus extra flags
# and prints steady-state traverse / shade launch times of one replayed round.
#   tools/replay.sh "<label>" "<extra hipcc flags>"
set -e
label="$1"; extra="$2"
make -s -B raytracing_amd/librtx.so EXTRA="-DRTX_REPLAY $extra" >/dev/null 2>&1
echo -n "$label: "
timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --spp 30 --steps 1 --warmup 0 2>&1 >/dev/null | grep -m1 REPLAY || true
