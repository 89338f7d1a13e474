#!/usr/bin/env python3
"""Prints the table of bench lines of one round from profiles/<tag>_bench_<config>.json (markdown).
    python3 tools/profiles_table.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = ["cover_spheres", "hello_triangle", "cornell_box", "quads", "simple_light", "simple_scene", "demo_spheres", "synthetic_1M_spheres", "synthetic_1M_triangles"]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    print("| Config | Frame | Kernel | Mrays/s | ms per frame | `roofline.bound`, `frac` | HBM measured frac | box / surface tests per ray | the reference's | lanes active | "
          "`image_check.max_rel_diff` | CPU Mrays/s |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for c in ORDER:
        path = os.path.join(ROOT, "profiles", f"{tag}_bench_{c}.json")
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        r = d["roofline"]
        v = r.get("valu_issue") or {}
        rw = r["reference_walk"]
        frame = d["metric"].split(", ", 1)[1].replace(c + " ", "")
        print(f"| {c} | {frame} | `{r['kernel']}` | {d['value']:.0f} | {d['ms_per_step']:.4g} | {r['bound']} {r['frac']:.3f} | {r['hbm']['measured_frac']:.3f} | "
              f"{r['nodes_per_ray']:.1f} / {r['sphere_tests_per_ray'] + r['planar_tests_per_ray']:.2f} | "
              f"{rw['nodes_per_ray']:.1f} / {rw['sphere_tests_per_ray'] + rw['planar_tests_per_ray']:.2f} | {v.get('lanes_active')} | "
              f"{d['image_check']['max_rel_diff']:.1e} | {d['cpu_baseline']['value']:.2f} |")


if __name__ == "__main__":
    main()
