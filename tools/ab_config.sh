#!/bin/bash
# A/B helper for the GPU box: rebuilds librtx.so with extra -D flags and renders one configuration of tools/config_sweep.py.
#   tools/ab_config.sh "<label>" "<extra hipcc flags>" <config name>
set -e
label="$1"; extra="$2"; config="$3"
make -s -B raytracing_amd/librtx.so EXTRA="$extra" >/dev/null 2>&1
timeout -k 10 300 python3 -u tools/config_sweep.py --only "$config" 2>/dev/null | LABEL="$label" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LABEL"], d["config"], d["mrays_per_s"], "Mrays/s", d["render_s"], "s")'
