#!/bin/bash
# Debug aid for the GPU box: builds with -DRTX_PROFILE_PHASES (+ extra flags) and prints where a traverse wave spends its cycles.
set -e
label="$1"; extra="$2"; shift 2
make -s -B raytracing_amd/librtx.so EXTRA="-DRTX_PROFILE_PHASES $extra" >/dev/null 2>&1
echo "== $label"
timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 30 "$@" 2>&1 >/dev/null | grep -E "PHASES|OCCUPANCY" | head -3
