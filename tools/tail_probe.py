#!/usr/bin/env python3
"""Times render calls of the headline scene at small sample counts: the cost of a call is
a + b * spp, and the intercept a (launch, top-up of an empty pool, drain of the last paths) is
what limits the speed-up when a frame's samples are split over many GPUs.

    python3 tools/tail_probe.py [--spps 1,2,4,8,16,31,62,125]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import raytracing_amd as rt  # noqa: E402
from raytracing_amd import scenes as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spps", default="1,2,4,8,16,31,62,125")
    args = ap.parse_args()
    surfaces, cam = S.cover_spheres(seed=0xC0FFEE, image_width=1920, samples_per_pixel=500)
    icam = rt.camera_initialize(cam)
    scene = rt.Scene(rt.HostBVH(surfaces))
    acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_accumulate(icam, 1, 0, 4, acc.data_ptr(), stream, 0)
    rows = []
    for spp in [int(x) for x in args.spps.split(",")]:
        ms = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = scene.render_accumulate(icam, 1, 0, spp, acc.data_ptr(), stream, 0)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        rows.append((spp, float(np.median(ms)), st.ms_total, st.iterations))
        print(json.dumps({"spp": spp, "call_ms_median": round(rows[-1][1], 3), "kernel_ms_last": round(st.ms_total, 3),
                          "rounds_of_busiest_wave": st.iterations}), flush=True)
    x = np.array([r[0] for r in rows[-4:]], float)
    y = np.array([r[1] for r in rows[-4:]], float)
    b, a = np.polyfit(x, y, 1)
    print(json.dumps({"fit_over_last_4": {"ms_per_spp": round(b, 4), "intercept_ms": round(a, 3)}}))


if __name__ == "__main__":
    main()
