#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
{ timeout -k 10 500 python3 tools/locality_probe.py --kind spheres 2>/dev/null; timeout -k 10 500 python3 tools/locality_probe.py --kind triangles 2>/dev/null; } | tee gpurun_out/r03/locality_probe.txt
