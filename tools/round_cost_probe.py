#!/usr/bin/env python3
"""Cost of a round (top-up, closest hit, shade) of the single-launch render when the sub-pools are nearly empty:
renders one sample per pixel of a tiny and of the full-size image at several depth limits and prints the kernel
time per round of the busiest wave.  The tiny image gives the latency of a round of one row of rays (about 11 us);
50 such rounds end every call whose longest path reaches the depth limit.

    python3 tools/round_cost_probe.py
"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import raytracing_amd as rt
from raytracing_amd import scenes as S
surfaces, cam = S.cover_spheres(seed=0xC0FFEE, image_width=1920, samples_per_pixel=500)
scene = rt.Scene(rt.HostBVH(surfaces))
stream = torch.cuda.current_stream().cuda_stream
for width, depth in ((64, 50), (64, 10), (64, 2), (64, 1), (1920, 1), (1920, 2), (1920, 10)):
    icam = rt.camera_initialize(dict(cam, image_width=width, max_depth=depth))
    acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    scene.render_accumulate(icam, 1, 0, 1, acc.data_ptr(), stream, 0)
    ms = [scene.render_accumulate(icam, 1, s, 1, acc.data_ptr(), stream, 0) for s in range(5)]
    best = min(ms, key=lambda s: s.ms_total)
    print(width, icam.image_height, "depth", depth, "kernel ms", round(best.ms_total, 4), "rounds", best.iterations, "rays", best.rays, "us/round", round(1e3 * best.ms_total / max(best.iterations, 1), 2))
