#!/bin/bash
# one GPU call: the new full-size test, then the atomics pricing probe (A/B of two prebuilt libraries)
set -e -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline" --timeout 200 --timeout-method thread 2>&1 | tail -5
timeout -k 10 600 bash tools/ab_so.sh 500 2 base noatom 2>&1 | tee gpurun_out/r03/atomics_probe.txt
cp abtmp/base.so raytracing_amd/librtx.so
