#!/bin/bash
# Runs on the GPU box: GPU tests, the headline bench (with its live counter passes, the CPU baseline and the cold frame), the bench line
# of every other configuration, a rocprofv3 kernel-trace of the headline bench command and the PMC passes.
#   tools/round_profile.sh TAG        results under gpurun_out/TAG; tools/collect_profiles.sh TAG copies the summaries into profiles/
# Every GPU step has its own time limit and the script stops at the first failure.
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== gpu tests"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > "$OUT/gputests.log" 2>&1 || { tail -40 "$OUT/gputests.log"; exit 1; }
tail -2 "$OUT/gputests.log"
echo "== bench (headline)"
timeout -k 10 500 python3 bench.py > "$OUT/bench_cover_spheres.json" 2> "$OUT/bench.err"
cat "$OUT/bench_cover_spheres.json"
echo "== bench (other configurations)"
for c in hello_triangle cornell_box quads simple_light simple_scene demo_spheres synthetic_1M_spheres synthetic_1M_triangles; do
  timeout -k 10 600 python3 bench.py --config $c --steps 2 > "$OUT/bench_$c.json" 2>> "$OUT/bench.err"
  python3 -c "
import json
d = json.load(open('$OUT/bench_$c.json')); r = d['roofline']
print('$c', d['value'], 'Mrays/s', d['ms_per_step'], 'ms;', r['kernel'], r['bound'], r['frac'], '; hbm measured', r['hbm']['measured_frac'], '; cpu', d['cpu_baseline']['value'], '; image check', d['image_check']['max_rel_diff'], d['image_check']['rays_equal'])"
done
echo "== rocprofv3 kernel trace of the headline bench command"
rm -rf "$OUT/stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
cat "$OUT/bench_under_rocprof.json"
find "$OUT/stats" -name '*kernel_trace.csv' -delete
# the stats file of THIS run (the directory was emptied above), kept under a fixed name; it must hold the kernel the line names
STATS=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
test -n "$STATS"
cp "$STATS" "$OUT/kernel_stats.csv"
python3 - "$OUT/bench_under_rocprof.json" "$OUT/kernel_stats.csv" <<'PY'
import csv, json, sys
line = json.load(open(sys.argv[1]))
kernel, launch_ms = line["roofline"]["kernel"], line["roofline"]["avg_launch_ms"]
rows = [r for r in csv.DictReader(open(sys.argv[2])) if kernel in r["Name"]]
assert rows, f"{sys.argv[2]} has no row for {kernel}, the kernel the bench line names"
row = max(rows, key=lambda r: float(r["TotalDurationNs"]))
avg_ms = float(row["AverageNs"]) / 1e6
print(f"kernel stats: {row['Name']}: {row['Calls']} calls, average {avg_ms:.3f} ms; the bench line's HIP events say {launch_ms:.3f} ms")
assert abs(avg_ms - launch_ms) <= 0.03 * launch_ms, "the profiler's average duration and the bench line's launch time disagree by more than 3 %"
PY
head -12 "$OUT/kernel_stats.csv"
echo "== pmc"
timeout -k 10 1000 python3 tools/pmc_profile.py "$OUT/pmc" --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-image-check --no-cold-frame > "$OUT/pmc.log" 2>&1
tail -30 "$OUT/pmc.log"
find "$OUT/pmc" -name '*counter_collection.csv' -delete
