#!/bin/bash
# Per-kernel times of the device BVH build of 10^6 surfaces (rocprofv3 kernel trace -> table on stdout).
# Run on the GPU box from the repository root: bash tools/build_profile.sh [out_dir]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$ROOT/gpurun_out/build_prof}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -o b1m -- python3 "$ROOT/tools/build_bench.py" --sizes 1000000 --no-host-above 0 > "$OUT/bench.log" 2>&1
grep '^{' "$OUT/bench.log"
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
q = "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, max(end-start)/1e3, max(vgpr_count) from kernels group by name order by 3 desc limit 24"
tot = 0.0
for r in c.execute(q):
    print(f"{r[0][:84]:84s} calls {r[1]:4d} total_ms {r[2]:7.3f} avg_us {r[3]:8.1f} max_us {r[4]:8.1f} vgpr {r[5]}")
    tot += r[2]
print("sum_ms", round(tot, 3))
PY
