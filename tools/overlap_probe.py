#!/usr/bin/env python3
"""Does running K independent render loops on K streams overlap traverse (VALU-bound) with
shade (latency-bound)?  K scenes, K host threads, each renders spp/K samples."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
from raytracing_amd import scenes as S

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
surfaces, cam = S.cover_spheres(seed=0xC0FFEE, image_width=1920, samples_per_pixel=spp)
bvh = rt.HostBVH(surfaces, rt.SAH_PER_SURFACE)
icam = rt.camera_initialize(cam)
dev = torch.device("cuda", 0)
acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device=dev)
for K in (1, 2, 3, 4):
    scenes = [rt.Scene(bvh) for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    def work(j, out):
        b = j * spp // K
        c = (j + 1) * spp // K - b
        out[j] = scenes[j].render_accumulate(icam, 7, b, c, acc.data_ptr(), streams[j].cuda_stream, 0)
    for rep in range(2):
        out = [None] * K
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(j, out)) for j in range(K)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rays = sum(o.rays for o in out)
    print(f"K={K}: {dt*1e3:.1f} ms, {rays/dt/1e6:.0f} Mrays/s", flush=True)
