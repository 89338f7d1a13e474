#!/bin/bash
# Runs on the GPU box: GPU tests, the headline bench (with its live counter passes and the CPU baseline), the bench line of every other
# configuration, a rocprofv3 kernel-trace of the headline bench command and the PMC passes.  Usage: tools/round_profile.sh TAG
# Every GPU step has its own time limit and the script stops at the first failure.
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== gpu tests"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/gputests.log" 2>&1
tail -2 "$OUT/gputests.log"
echo "== bench (headline)"
timeout -k 10 400 python3 bench.py > "$OUT/bench_cover_spheres.json" 2> "$OUT/bench.err"
cat "$OUT/bench_cover_spheres.json"
echo "== bench (other configurations)"
for c in hello_triangle cornell_box quads simple_light simple_scene demo_spheres synthetic_1M_spheres synthetic_1M_triangles; do
  timeout -k 10 600 python3 bench.py --config $c --steps 2 > "$OUT/bench_$c.json" 2>> "$OUT/bench.err"
  python3 -c "
import json
d = json.load(open('$OUT/bench_$c.json')); r = d['roofline']
print('$c', d['value'], 'Mrays/s', d['ms_per_step'], 'ms;', r['bound'], r['frac'], '; hbm measured', r['hbm']['measured_frac'], '; cpu', d['cpu_baseline']['value'])"
done
echo "== rocprofv3 kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
cat "$OUT/bench_under_rocprof.json"
find "$OUT/stats" -name '*kernel_trace.csv' -delete
find "$OUT/stats" -name '*kernel_stats.csv' -exec head -12 {} \;
echo "== pmc"
timeout -k 10 1000 python3 tools/pmc_profile.py "$OUT/pmc" --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > "$OUT/pmc.log" 2>&1
tail -30 "$OUT/pmc.log"
find "$OUT/pmc" -name '*counter_collection.csv' -delete
