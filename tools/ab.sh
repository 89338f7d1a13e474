#!/bin/bash
# A/B helper for the GPU box: rebuilds librtx.so with extra -D flags and runs a short bench.
#   tools/ab.sh "<label>" "<extra hipcc flags>" [bench args...]
set -e
label="$1"; extra="$2"; shift 2
make -s -B raytracing_amd/librtx.so EXTRA="$extra" >/dev/null 2>&1
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline "$@" 2>/dev/null | LABEL="$label" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LABEL"], d["value"], "Mrays/s", d["ms_per_step"], "ms")'
