"""Known answers for the device path that do not go through the CPU oracle.

The oracle and the kernels were written by reading the same source text, so a shared misreading of a material would
pass every oracle comparison.  These scenes have radiance values that follow from the reference's formulas by hand
(src/material.rs:20-137, src/camera.rs:184-208); the device render is compared with those numbers directly.
No reference-produced fixture covers Metal, Dielectric or sphere shading ("parity unpinned" for them): this is the
independent check there is.
"""
import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu


def render(rt, surfaces, cam, seed=3):
    img, st = rt.Scene(rt.HostBVH(S.pack(surfaces))).render(rt.camera_initialize(cam), seed)
    return img, st


def test_lambertian_sphere_in_a_white_furnace(rt):
    """A convex Lambertian surface under a uniform background B: the scattered direction n + unit vector points
    away from the sphere, so every path is hit -> scatter -> miss and carries albedo * B (camera.rs:197-207).
    Pixels wholly on the sphere equal albedo * B, pixels wholly off it equal B, whatever the random numbers were."""
    albedo, back = (0.25, 0.5, 0.75), (1.0, 0.5, 0.125)
    cam = S.camera(aspect_ratio=1.0, image_width=128, samples_per_pixel=32, max_depth=50, v_fov=40.0,
                   look_from=(0.0, 0.0, 6.0), look_at=(0.0, 0.0, 0.0), background=back)
    img, st = render(rt, [S.sphere((0.0, 0.0, 0.0), 1.5, S.lambertian(albedo))], cam)
    want_on = np.array(albedo) * np.array(back)
    assert np.allclose(img[64, 64], want_on, rtol=1e-13, atol=0)
    assert np.allclose(img[40:88, 40:88], want_on, rtol=1e-13, atol=0)           # the sphere covers the middle of the frame
    assert np.allclose(img[:4, :4], back, rtol=1e-13, atol=0)                    # the corners see only the background
    on = np.isclose(img, want_on, rtol=1e-13).all(axis=-1)
    off = np.isclose(img, back, rtol=1e-13).all(axis=-1)
    # silhouette: pi r'^2 pixels, r' = projected radius; everything that is neither is a mix on the silhouette
    assert 0.9 < on.sum() / (np.pi * (1.5 / np.sqrt(36 - 2.25) / np.tan(np.radians(20)) * 64) ** 2) < 1.02
    assert (~on & ~off).sum() < 0.05 * img.shape[0] * img.shape[1]
    assert st.paths < st.rays < 2 * st.paths


def test_glass_sphere_in_a_white_furnace_conserves_energy(rt):
    """Dielectric attenuation is (1, 1, 1) (material.rs:125-128) and it always scatters: under background (1, 1, 1)
    every sample returns exactly 1 unless the path is cut at max_depth (then 0, camera.rs:190-192).  So spp * pixel is
    an integer, never above spp, and with 50 bounces allowed almost every sample arrives."""
    spp = 16
    cam = S.camera(aspect_ratio=1.0, image_width=160, samples_per_pixel=spp, max_depth=50, v_fov=30.0,
                   look_from=(0.0, 0.5, 5.0), look_at=(0.0, 0.0, 0.0), background=(1.0, 1.0, 1.0))
    # (the bubble is nudged off centre: coincident centroids have no SAH splitting plane, as in examples/demo_spheres.rs:65)
    glass = [S.sphere((0.0, 0.0, 0.0), 1.0, S.dielectric(1.5)), S.sphere((0.0, 0.0, 1e-4), 0.6, S.dielectric(1.0 / 1.5))]
    img, st = render(rt, glass, cam)
    counts = img * spp
    assert np.allclose(counts, np.round(counts), atol=1e-9) and counts.max() <= spp + 1e-9 and counts.min() >= 0
    assert img.mean() > 0.999
    assert st.rays > 2 * st.paths          # the middle of the frame refracts through four interfaces
    # cut at the first interface instead: pixels on the glass lose everything, pixels off it nothing
    dark, _ = render(rt, glass, dict(cam, max_depth=1))
    assert np.array_equal(dark[80, 80], [0.0, 0.0, 0.0]) and np.array_equal(dark[0, 0], [1.0, 1.0, 1.0])


def test_mirror_shows_the_emitter_behind_the_camera(rt):
    """Metal with fuzz 0 reflects exactly (material.rs:84-96: reflect(d, n).to_unit() + 0 * unit vector).  A mirror quad
    in the plane z = 0 seen head-on from z = 5, and a two-sided emitter (material.rs:41) in the plane z = 10 behind
    the camera: every path is mirror -> light, radiance = albedo * emit; a pixel beside the mirror sees the background."""
    albedo, emit = (0.8, 0.6, 0.4), (2.0, 3.0, 5.0)
    cam = S.camera(aspect_ratio=1.0, image_width=96, samples_per_pixel=8, max_depth=10, v_fov=60.0,
                   look_from=(0.0, 0.0, 5.0), look_at=(0.0, 0.0, 0.0), background=(0.0, 0.0, 0.25))
    scene = [S.quad((-2.0, -2.0, 0.0), (4.0, 0.0, 0.0), (0.0, 4.0, 0.0), S.metal(albedo, 0.0)),
             S.quad((-50.0, -50.0, 10.0), (100.0, 0.0, 0.0), (0.0, 100.0, 0.0), S.diffuse_light(emit))]
    img, st = render(rt, scene, cam)
    want = np.array(albedo) * np.array(emit)
    assert np.allclose(img[30:66, 30:66], want, rtol=1e-13, atol=0)     # |x|, |y| < 1.1 at z = 0: well inside the mirror
    assert np.allclose(img[0, 0], [0.0, 0.0, 0.25], rtol=0, atol=0)     # (-2.9, 2.9) at z = 0: beside it
    # a fuzzed mirror still reflects upwards from this side (|fuzz| < 1 keeps dot(fuzzed, n) > 0 for head-on rays) and the
    # emitter covers the whole upper half space in practice, so the centre keeps its value
    fuzzy, _ = render(rt, [S.quad((-2.0, -2.0, 0.0), (4.0, 0.0, 0.0), (0.0, 4.0, 0.0), S.metal(albedo, 0.3)), scene[1]], cam)
    assert np.allclose(fuzzy[44:52, 44:52], want, rtol=1e-13, atol=0)


def test_schlick_reflectance_at_normal_incidence(rt):
    """A thick glass slab cannot be built from the reference's surfaces, but one interface can be measured: a huge glass
    sphere seen head-on.  At normal incidence cos = 1, Schlick's r0 = ((1 - 1/1.5) / (1 + 1/1.5))^2 = 0.04
    (material.rs:134-137): a fraction r0 of the samples reflects straight back into a black background half and the rest
    enters the glass.  With max_depth 1 nothing returns at all; with max_depth 2 exactly the reflected samples return the
    background behind the camera.  The reflected fraction is binomial(n, 0.04)."""
    n_side, spp = 64, 64
    cam = S.camera(aspect_ratio=1.0, image_width=n_side, samples_per_pixel=spp, max_depth=2, v_fov=0.02,
                   look_from=(0.0, 0.0, 2000.0), look_at=(0.0, 0.0, 0.0), background=(1.0, 1.0, 1.0))
    img, st = render(rt, [S.sphere((0.0, 0.0, 0.0), 1000.0, S.dielectric(1.5))], cam)
    # depth 2: reflected samples miss everything on the way back and collect the background; refracted ones hit the
    # far side of the sphere from inside and are cut there
    frac = img[..., 0].mean()
    n = n_side * n_side * spp
    sigma = np.sqrt(0.04 * 0.96 / n)
    assert abs(frac - 0.04) < 5 * sigma, (frac, sigma)
