#!/usr/bin/env python3
"""One rocprofv3 --pmc pass with the counters given on the command line:
    python3 tools/pmc_quick.py OUTDIR COUNTER[,COUNTER...] [bench args... | SCRIPT.py args...]
Profiles bench.py, or another script when the first extra argument ends in .py.  Prints per-kernel totals."""
import csv, glob, os, subprocess, sys
out, ctrs = sys.argv[1], sys.argv[2].split(",")
bench_args = sys.argv[3:] or ["--spp", "60", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
script = ["bench.py"] if not bench_args[0].endswith(".py") else []
cmd = ["timeout", "-s", "KILL", "240", "rocprofv3", "--pmc", *ctrs, "--output-format", "csv", "-d", out, "--", "python3", *script, *bench_args]
r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
if r.returncode != 0:
    print("failed", r.returncode, r.stdout[-600:]); sys.exit(1)
tot = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rtx::", "")
        tot.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in tot.items():
    if k.startswith("k_"):
        print(k, {c: f"{x:.4g}" for c, x in sorted(v.items())})
