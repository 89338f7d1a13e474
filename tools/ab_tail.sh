#!/bin/bash
# A/B helper for the GPU box: rebuilds librtx.so with extra -D flags and times render calls at several sample counts.
#   tools/ab_tail.sh "<label>" "<extra hipcc flags>" [tail_probe args...]
set -e
label="$1"; extra="$2"; shift 2
make -s -B raytracing_amd/librtx.so EXTRA="$extra" >/dev/null 2>&1
echo "== $label"
timeout -k 10 200 python3 tools/tail_probe.py "$@" 2>/dev/null
