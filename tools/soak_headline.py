"""Soak check for the GPU box: the headline frame rendered 60 times; every frame must trace the same number of rays and sum to the same image (up to the order of the f64 additions)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import raytracing_amd as rt
from raytracing_amd import scenes as S
surfaces, cam = S.cover_spheres()
cam = dict(cam, image_width=1920, samples_per_pixel=500)
bvh = rt.HostBVH(surfaces)
scene = rt.Scene(bvh)
icam = rt.camera_initialize(cam)
acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
rays, sums, forms = [], [], []
t0 = time.time()
for i in range(60):
    acc.zero_()
    st = scene.render_accumulate(icam, 12648430, 0, 500, acc.data_ptr(), stream)
    rays.append(int(st.rays)); sums.append(float(acc.sum().item())); forms.append(int(st.kernel_form))
    if i % 10 == 0: print(i, rays[-1], sums[-1], forms[-1], round(st.ms_total, 2), flush=True)
print("frames", len(rays), "distinct ray counts", sorted(set(rays)), "forms", sorted(set(forms)))
rel = (max(sums[1:]) - min(sums[1:])) / abs(sums[1])
print("checksum spread (relative)", rel, "seconds", round(time.time() - t0, 1))
assert len(set(rays)) == 1 and rel < 1e-11
print("SOAK OK")
