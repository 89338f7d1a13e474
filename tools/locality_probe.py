#!/usr/bin/env python3
"""How the rate of the global-memory walk depends on the size of the tree it wanders over: sphere fields of the density
of the 10^6-sphere configuration (one sphere per unit volume) with n spheres in a cube of side 100 (n / 10^6)^(1/3),
camera pulled back in proportion.  A field of 10^6 / 8 spheres is what each of the eight XCDs' L2s would see if rays were
shared out among the XCDs by where they are.
    python3 tools/locality_probe.py [--sizes 31250,125000,500000,1000000] [--kind spheres|triangles]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="31250,62500,125000,250000,500000,1000000")
    ap.add_argument("--kind", default="spheres")
    ap.add_argument("--spp", type=int, default=4)
    args = ap.parse_args()
    import torch
    import raytracing_amd as rt
    from raytracing_amd import scenes as S

    for n in [int(x) for x in args.sizes.split(",")]:
        s = (n / 1e6) ** (1.0 / 3.0)
        surfaces, cam = S.synthetic_soup(n=n, kind=args.kind, seed=42, image_width=3840, samples_per_pixel=args.spp)
        surfaces["q"] *= s
        cam = dict(cam, look_from=(0.0, 0.0, -160.0 * s))
        icam = rt.camera_initialize(cam)
        bvh = rt.HostBVH(surfaces, rt.SAH_PER_SURFACE, device=True)
        scene = rt.Scene(bvh)
        acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
        st_ = torch.cuda.current_stream().cuda_stream
        scene.render_accumulate(icam, 1, 0, 1, acc.data_ptr(), st_)
        st = scene.render_accumulate(icam, 1, 1, args.spp, acc.data_ptr(), st_)
        cnt = scene.render_accumulate(icam, 1, 1, 1, acc.data_ptr(), st_, rt.FLAG_COUNT_ACTUAL)
        print(json.dumps(dict(kind=args.kind, n=n, nodes_MB=round(bvh.n_nodes * 16 / 1e6, 1), ms=round(st.ms_total, 2), mrays=round(st.rays / st.ms_total / 1e3, 1),
                              rays_per_path=round(st.rays / st.paths, 2), nodes_per_ray=round(cnt.node_visits / cnt.rays, 1),
                              surface_tests_per_ray=round((cnt.sphere_tests + cnt.planar_tests) / cnt.rays, 2), form=rt.FORM_KERNEL.get(st.kernel_form))), flush=True)
        scene.close()
        bvh.close()


if __name__ == "__main__":
    main()
