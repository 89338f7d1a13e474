"""Device time of a scene's FIRST render call (what a one-frame-per-process caller sees) with and without the first-sample probe.
Usage: python3 tools/first_call_probe.py [config] [spp]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402
from raytracing_amd import scenes as S  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cover_spheres"
surfaces, cam = getattr(S, name)()
spp = int(sys.argv[2]) if len(sys.argv) > 2 else cam["samples_per_pixel"]
icam = rt.camera_initialize(cam)
stream = torch.cuda.current_stream().cuda_stream
acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
warm = rt.Scene(rt.HostBVH(surfaces))
warm.render_accumulate(icam, 1, 0, 2, acc.data_ptr(), stream)   # code objects loaded, clocks up
for probe in ("0", "1", "0", "1"):
    os.environ["RTX_PROBE_FIRST_CALL"] = probe
    dev = rt.Scene(rt.HostBVH(surfaces))
    acc.zero_()
    st = dev.render_accumulate(icam, 1, 0, spp, acc.data_ptr(), stream)
    print(json.dumps(dict(config=name, spp=spp, probe=probe, first_call_ms=round(st.ms_total, 3), mrays=round(st.rays / st.ms_total / 1e3, 1))))
    dev.close()
