#!/usr/bin/env python3
"""Collects rocprofv3 PMC counters for the render kernels, one pass per
counter group (gfx950 holds at most 8 SQ / 4 TCC / 4 TCP / 2 TA / 2 GRBM
counters per run), and writes per-kernel totals to a JSON + markdown summary.

    python3 tools/pmc_profile.py OUTDIR [bench.py args...]

Runs on the GPU box.  Each pass is its own rocprofv3 process under a hard
timeout; the profiled program is `python3 bench.py ...` directly.
"""
import csv
import glob
import json
import os
import subprocess
import sys

PASSES = [
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"],
    ["SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"],
    ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_BRANCH"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TA_BUSY_avr", "GRBM_GUI_ACTIVE"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_ATOMIC_sum", "TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum"],
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_REQ_sum", "TCP_TOTAL_ACCESSES_sum", "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("rtx::", "").strip()


def main():
    out = sys.argv[1]
    bench_args = sys.argv[2:] or ["--spp", "10", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-pmc"]
    os.makedirs(out, exist_ok=True)
    totals, calls = {}, {}
    for i, ctrs in enumerate(PASSES):
        d = os.path.join(out, f"pass{i}")
        cmd = ["timeout", "-s", "KILL", "180", "rocprofv3", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "--",
               "python3", "bench.py", *bench_args]
        print("pass", i, " ".join(ctrs), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print("  pass failed rc", r.returncode, r.stdout[-400:], flush=True)
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    totals.setdefault(k, {}).setdefault(c, 0.0)
                    totals[k][c] += float(row["Counter_Value"])
                    key = (k, row["Dispatch_Id"])
                    if c == ctrs[0] and key not in seen:
                        seen.add(key)
                        calls.setdefault(k, {})[i] = calls.setdefault(k, {}).get(i, 0) + 1
    summary = {k: dict(v, dispatches=max(calls.get(k, {0: 0}).values())) for k, v in totals.items()}
    with open(os.path.join(out, "pmc_summary.json"), "w") as fh:
        json.dump({"bench_args": bench_args, "kernels": summary}, fh, indent=1, sort_keys=True)
    keep = [k for k in summary if k.startswith("k_")]
    ctr_names = sorted({c for k in keep for c in summary[k]})
    with open(os.path.join(out, "pmc_summary.md"), "w") as fh:
        fh.write("bench args: " + " ".join(bench_args) + "\n\n| counter | " + " | ".join(keep) + " |\n|---|" + "---|" * len(keep) + "\n")
        for c in ctr_names:
            fh.write(f"| {c} | " + " | ".join(f"{summary[k].get(c, 0):.4g}" for k in keep) + " |\n")
    print(open(os.path.join(out, "pmc_summary.md")).read())
    # HBM traffic per launch.  FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte
    # read requests as 64 bytes, so reads are doubled (an upper bound where narrower requests are mixed in).
    traffic = {}
    for k in keep:
        n = summary[k].get("dispatches", 0)
        if n and "FETCH_SIZE" in summary[k] and "WRITE_SIZE" in summary[k]:
            rd = 2.0 * summary[k]["FETCH_SIZE"] * 1024.0 / n
            wr = summary[k]["WRITE_SIZE"] * 1024.0 / n
            traffic[k] = {"hbm_bytes_per_launch": round(rd + wr, 1), "read_bytes_per_launch": round(rd, 1),
                          "write_bytes_per_launch": round(wr, 1), "launches": n,
                          "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python3 bench.py " + " ".join(bench_args)}
    with open(os.path.join(out, "hbm_traffic.json"), "w") as fh:
        json.dump({"kernels": traffic}, fh, indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
