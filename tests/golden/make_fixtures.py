"""Writes the generated fixtures of tests/golden/ (run from the repo root:
`python tests/golden/make_fixtures.py`).  Uses the CPU oracle only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from raytracing_amd import scenes as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    surfaces, _ = S.cover_spheres()
    np.savez_compressed(os.path.join(HERE, "cover_spheres_seed_c0ffee.npz"), surfaces=surfaces)

    # closest-hit vectors: primary rays of a small camera plus rays leaving the
    # primary hit points in seeded random directions, per example scene
    out = {}
    for name, make in S.EXAMPLES.items():
        surf, cam = make()
        sc = po.Scene(surf)
        ic = po.camera_initialize(dict(cam, image_width=96))
        org, d = po.primary_rays(ic, seed=2024, sample=0)
        ids, t, _ = sc.trace_closest(org, d)
        hit = ids != po.NONE
        rng = np.random.default_rng(5)
        o2 = org[hit] + d[hit] * t[hit, None]
        d2 = rng.normal(size=o2.shape)
        O, D = np.concatenate([org, o2]), np.concatenate([d, d2])
        ids, t, ctr = sc.trace_closest(O, D)
        out[name + "_org"], out[name + "_dir"] = O, D
        out[name + "_id"], out[name + "_t"] = ids, t
        out[name + "_counts"] = np.array([ctr.node_visits, ctr.sphere_tests, ctr.planar_tests], np.uint64)
    np.savez_compressed(os.path.join(HERE, "trace_vectors.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.endswith("_id")})


if __name__ == "__main__":
    main()
