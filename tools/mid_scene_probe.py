#!/usr/bin/env python3
"""Render time of mid-size sphere scenes with the tree in LDS (one 1024-thread block per CU) and in global memory:
    python3 tools/mid_scene_probe.py [--sizes 600,1500,2300]"""
import argparse, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(n, wide):
    code = f"""
import os, sys, json, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import raytracing_amd as rt
from raytracing_amd import scenes as S
surfaces, cam = S.synthetic_soup(n={n}, kind="spheres", seed=7, image_width=1920, samples_per_pixel=16)
icam = rt.camera_initialize(cam)
scene = rt.Scene(rt.HostBVH(surfaces))
acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
scene.render_accumulate(icam, 1, 0, 2, acc.data_ptr(), s)
st = scene.render_accumulate(icam, 1, 0, 16, acc.data_ptr(), s)
print(json.dumps(dict(n={n}, wide={wide!r}, ms=round(st.ms_total, 2), mrays=round(st.rays / st.ms_total / 1e3, 1))))
"""
    env = dict(os.environ, RTX_LDS_WIDE_BYTES=str(150 * 1024) if wide else "0")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else f"failed n={n} wide={wide} rc={r.returncode}", flush=True)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="600,1500,2300")
    for n in [int(x) for x in ap.parse_args().sizes.split(",")]:
        run(n, True)
        run(n, False)
