#!/bin/bash
# A/B helper for the GPU box: renders configurations of tools/config_sweep.py under different values of one environment variable.
#   tools/ab_env.sh VAR "v1 v2 ..." config1 config2 ...
set -e
var="$1"; vals="$2"; shift 2
for c in "$@"; do
  for v in $vals; do
    env "$var=$v" timeout -k 10 300 python3 -u tools/config_sweep.py --only "$c" 2>/dev/null | LABEL="$var=$v" python3 -c '
import json, os, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["LABEL"], d["config"], d["mrays_per_s"], "Mrays/s", d["render_s"], "s", flush=True)'
  done
done
