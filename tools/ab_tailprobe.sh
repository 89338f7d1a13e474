#!/bin/bash
# A/B helper for the GPU box: the cost of small render calls (tools/tail_probe.py) with each of the prebuilt libraries
# given (abtmp/<name>.so).   tools/ab_tailprobe.sh name1 name2 ...
set -e
for n in "$@"; do
  cp "abtmp/$n.so" raytracing_amd/librtx.so
  echo "== $n"
  timeout -k 10 300 python3 -u tools/tail_probe.py --spps 1,4,16,62,125,250
done
