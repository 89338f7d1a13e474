import os, sys, json, subprocess
ROOT = "/root/repo" if os.path.isdir("/root/repo/raytracing_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
code = """
import os, sys, json, torch
sys.path.insert(0, %r)
import raytracing_amd as rt
from raytracing_amd import scenes as S
n, kind, spp = %d, %r, %d
if kind == 'cover':
    surfaces, cam = S.cover_spheres(seed=0xC0FFEE, image_width=1920, samples_per_pixel=spp)
else:
    surfaces, cam = S.synthetic_soup(n=n, kind=kind, seed=7, image_width=1920, samples_per_pixel=spp)
icam = rt.camera_initialize(cam)
scene = rt.Scene(rt.HostBVH(surfaces))
acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device='cuda')
s = torch.cuda.current_stream().cuda_stream
scene.render_accumulate(icam, 1, 0, 2, acc.data_ptr(), s)
best = None
for _ in range(3):
    st = scene.render_accumulate(icam, 1, 0, spp, acc.data_ptr(), s)
    best = st.ms_total if best is None else min(best, st.ms_total)
print(json.dumps(dict(n=n, kind=kind, spp=spp, resident=os.environ.get('RTX_RESIDENT'), ms=round(best, 3), rays_per_path=round(st.rays / max(st.paths, 1), 2), mrays=round(st.rays / best / 1e3, 1))))
"""
for (n, kind, spp) in [(600, 'spheres', 16), (600, 'spheres', 64), (300, 'spheres', 32), (0, 'cover', 32), (400, 'triangles', 32)]:
    for res in ("0", "1"):
        env = dict(os.environ, RTX_RESIDENT=res, RTX_BOUNCE_AT="52")
        r = subprocess.run([sys.executable, "-c", code % (ROOT, n, kind, spp)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "failed %s" % r.returncode, flush=True)
