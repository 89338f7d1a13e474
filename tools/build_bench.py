#!/usr/bin/env python3
"""Times BVH::from_slice on the device (rtx_bvh_build) against the host builder
on random soups and checks that both return the same node array.

    python3 tools/build_bench.py [--sizes 10000,100000,1000000] [--strategy 3] [--no-host-above N]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import raytracing_amd as rt  # noqa: E402
from raytracing_amd import scenes as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="10000,100000,1000000")
    ap.add_argument("--strategy", type=int, default=rt.SAH_PER_SURFACE)
    ap.add_argument("--buckets", type=int, default=16)
    ap.add_argument("--no-host-above", type=int, default=2_000_000, help="skip the host build (and the comparison) above this size")
    args = ap.parse_args()
    rt.HostBVH(S.synthetic_soup(n=64, kind="spheres")[0], args.strategy, args.buckets, device=True).close()   # code loaded, context up
    for n in [int(x) for x in args.sizes.split(",")]:
        for kind in ("spheres", "triangles"):
            surfaces, _ = S.synthetic_soup(n=n, kind=kind, seed=42)
            t0 = time.perf_counter()
            dev = rt.HostBVH(surfaces, args.strategy, args.buckets, device=True)
            dev_wall = time.perf_counter() - t0
            st = dev.build_stats
            row = {"surfaces": n, "kind": kind, "strategy": args.strategy, "levels": st.levels, "device_sort_ms": round(st.ms_sort, 3),
                   "device_levels_ms": round(st.ms_levels, 3), "device_build_ms": round(st.ms_sort + st.ms_levels, 3),
                   "device_mb": round(st.device_bytes / 2**20, 1), "call_wall_s_incl_host_tables": round(dev_wall, 3)}
            if n <= args.no_host_above:
                t0 = time.perf_counter()
                host = rt.HostBVH(surfaces, args.strategy, args.buckets)
                row["host_build_s"] = round(time.perf_counter() - t0, 3)
                a, b = dev.node_tables(), host.node_tables()
                row["identical"] = all(np.array_equal(x.view(np.uint64) if x.dtype == np.float64 else x,
                                                      y.view(np.uint64) if y.dtype == np.float64 else y) for x, y in zip(a, b))
                row["host_over_device"] = round(row["host_build_s"] * 1e3 / row["device_build_ms"], 1)
                host.close()
            dev.close()
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
