#!/bin/bash
# Runs on the GPU box: the gather micro-benchmark, then two rocprofv3 counter passes over its 4 GiB / 8 waves per SIMD case
# to see how many bytes the fabric delivers per scattered read.  Usage: tools/ubench/gather_pmc.sh
set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/gather
mkdir -p "$OUT"
BIN=tools/ubench/gather_rate
timeout -k 10 120 $BIN > "$OUT/rates.txt"
cat "$OUT/rates.txt"
for REC in 32 64 128; do
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch$REC" -- $BIN 4096 8 $REC > /dev/null
  timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d "$OUT/req$REC" -- $BIN 4096 8 $REC > /dev/null
done
python3 - <<'PY'
import csv, glob
for rec in (32, 64, 128):
    for kind in ("fetch", "req"):
        for f in glob.glob(f"gpurun_out/gather/{kind}{rec}/**/*counter_collection.csv", recursive=True):
            rows = {}
            for r in csv.DictReader(open(f)):
                rows.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:40], r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
            for k, v in sorted(rows.items(), key=lambda kv: int(kv[0][0])):
                print(rec, k, v)
PY
