#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_cull_boxes.py -m gpu -x -q -s --timeout 200 --timeout-method thread 2>&1 | grep -v '^$' | tail -40
