#!/usr/bin/env python3
"""Renders every benchmark configuration on one GPU and prints one JSON line
per configuration: wall time of the render call, Mrays/s, rays per path and
the work counters per ray.  Configurations that would take minutes at their
nominal sample count are run at a reduced count (cost is linear in spp) and
say so in `spp_run`.

    python3 tools/config_sweep.py [--full] [--only NAME]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import raytracing_amd as rt  # noqa: E402
from raytracing_amd import scenes as S  # noqa: E402


def configs(full):
    yield "hello_triangle", S.hello_triangle(), rt.SAH_PER_SURFACE, None
    sp = S.cover_spheres(seed=0xC0FFEE, image_width=1920, samples_per_pixel=500)
    yield "cover_spheres_1080p", sp, rt.SAH_PER_SURFACE, None
    surf, cam = S.cornell_box()
    yield "cornell_box_20k", (surf, dict(cam, samples_per_pixel=20000)), rt.SAH_PER_SURFACE, None if full else 2000
    yield "quads", S.quads(), rt.SAH_PER_SURFACE, None
    yield "simple_light", S.simple_light(), rt.SAH_PER_SURFACE, None
    yield "simple_scene", S.simple_scene(), rt.SAH_PER_SURFACE, None
    yield "demo_spheres", S.demo_spheres(), rt.SAH_PER_SURFACE, None
    for kind in ("spheres", "triangles"):
        yield f"synthetic_1M_{kind}_4k", S.synthetic_soup(n=1_000_000, kind=kind), rt.SAH_PER_SURFACE, 1000 if full else 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="nominal sample counts everywhere")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for name, (surfaces, cam), strategy, spp_run in configs(args.full):
        if args.only and args.only != name:
            continue
        print(f"# {name}: building", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        bvh = rt.HostBVH(surfaces, strategy, device=True)   # the tree of the host builder, partitioned on the device
        build_s = time.perf_counter() - t0
        bst = bvh.build_stats
        icam = rt.camera_initialize(cam)
        t0 = time.perf_counter()
        scene = rt.Scene(bvh)
        scene_s = time.perf_counter() - t0
        spp_nominal = icam.samples_per_pixel
        spp = spp_run or spp_nominal
        acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device=dev)
        print(f"# {name}: rendering {spp} spp", file=sys.stderr, flush=True)
        warm = min(spp, 2)
        scene.render_accumulate(icam, 1, 0, warm, acc.data_ptr(), stream, 0)   # pools allocated, code loaded
        acc.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = scene.render_accumulate(icam, 1, 0, spp, acc.data_ptr(), stream, 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cnt = scene.render_accumulate(icam, 1, 0, min(spp, 2), acc.data_ptr(), stream, rt.FLAG_COUNT_WORK)
        mean = float(acc.mean().item())
        print(json.dumps({
            "config": name, "image": f"{icam.image_width}x{icam.image_height}", "spp_nominal": spp_nominal, "spp_run": spp,
            "surfaces": int(len(surfaces)), "bvh_nodes": bvh.n_nodes, "bvh_build_call_s": round(build_s, 3),
            "bvh_build_device_ms": round(bst.ms_sort + bst.ms_levels, 3), "bvh_levels": bst.levels, "scene_create_s": round(scene_s, 3),
            "render_s": round(dt, 4), "render_s_at_nominal_spp": round(dt * spp_nominal / spp, 3),
            "paths": st.paths, "rays": st.rays, "rays_per_path": round(st.rays / st.paths, 3),
            "mrays_per_s": round(st.rays / dt / 1e6, 1), "rounds": st.iterations,
            "nodes_per_ray": round(cnt.node_visits / cnt.rays, 2), "sphere_tests_per_ray": round(cnt.sphere_tests / cnt.rays, 3),
            "planar_tests_per_ray": round(cnt.planar_tests / cnt.rays, 3),
            "box_tests_left_to_f64": cnt.box_fallbacks / max(cnt.node_visits, 1), "image_mean": mean,
        }), flush=True)
        scene.close()
        bvh.close()


if __name__ == "__main__":
    main()
