"""Parity of the HIP path against the CPU oracle, through the C ABI.

Bars: closest-hit surface ids and hit distances bit-exact; primary rays
bit-exact; per-sample radiance bit-exact against the oracle's loop form of
ray_color and within 1e-12 relative of its recursive (as-written) form;
converged images against the reference's PNG within the stated RMSE.
"""
import os

import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NONE = 0xFFFFFFFF
EXAMPLES = list(S.EXAMPLES)


@pytest.fixture(scope="module")
def built(rt, oracle):
    """name -> (surfaces, camera dict, oracle scene, device scene)"""
    assert rt.device_count() >= 1, "no HIP device visible"
    cache = {}

    def get(name):
        if name not in cache:
            surfaces, cam = S.EXAMPLES[name]()
            cache[name] = (surfaces, cam, oracle.Scene(surfaces), rt.Scene(rt.HostBVH(surfaces)))
        return cache[name]

    return get


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


# ------------------------------------------------------------------ arithmetic the parity rests on
def test_device_division_and_sqrt_are_correctly_rounded(rt):
    rng = np.random.default_rng(0)
    n = 1 << 20
    a = rng.normal(size=n) * 10.0 ** rng.integers(-30, 30, n)
    b = rng.normal(size=n) * 10.0 ** rng.integers(-30, 30, n)
    edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 1.7976931348623157e308, 2.2250738585072014e-308, 3.0, 1e-160])
    a[: edge.size * edge.size] = np.repeat(edge, edge.size)
    b[: edge.size * edge.size] = np.tile(edge, edge.size)
    with np.errstate(all="ignore"):
        want_div, want_sqrt = a / b, np.sqrt(np.abs(a))
    got_div = rt.selftest_math(0, a, b)
    got_sqrt = rt.selftest_math(1, np.abs(a), b)
    nan = np.isnan(want_div)
    assert np.array_equal(np.isnan(got_div), nan)
    assert np.array_equal(bits(got_div)[~nan], bits(want_div)[~nan])
    nan = np.isnan(want_sqrt)
    assert np.array_equal(bits(got_sqrt)[~nan], bits(want_sqrt)[~nan])


def test_device_f32_division_is_correctly_rounded(rt):
    # the single-precision tier of the box test takes inv32 = 1.0f / RN32(d) to be within 2^-24 of the quotient
    rng = np.random.default_rng(5)
    n = 1 << 20
    a = np.ones(n)
    a[n // 2:] = rng.normal(size=n - n // 2) * 10.0 ** rng.integers(-15, 15, n - n // 2)
    b = rng.normal(size=n) * 10.0 ** rng.integers(-30, 30, n)
    b[:6] = [3.0, -7.0, 1e-30, 1e30, 0.1, 1.0]
    with np.errstate(all="ignore"):
        want = (a.astype(np.float32) / b.astype(np.float32)).astype(np.float64)
    got = rt.selftest_math(3, a, b)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(bits(got)[~nan], bits(want)[~nan])


def test_device_f32_reciprocal_of_the_box_tier_is_within_its_bound(rt):
    # the single-precision tier of the box test takes inv32 = recip32(RN32(d)) (approximate reciprocal + one Newton step)
    # to be within (1 + 2^-18) 2^-24 of the reciprocal, for the rays it admits: 1e-30 <= |inv32| <= 1e15
    rng = np.random.default_rng(6)
    n = 1 << 21
    b = rng.normal(size=n) * 10.0 ** rng.uniform(-15, 30, n)
    b[: n // 4] = rng.uniform(-4.0, 4.0, n // 4)          # where a path tracer's directions live
    b[:8] = [3.0, -7.0, 1e-15, 1e30, 0.1, 1.0, -1.0, 2.0 ** -20]
    x32 = b.astype(np.float32)
    exact = 1.0 / x32.astype(np.float64)
    keep = (np.abs(exact) >= 1e-30) & (np.abs(exact) <= 1e15)
    got = rt.selftest_math(4, np.ones(n), b)
    assert np.all(np.abs(got - exact)[keep] <= (1 + 2.0 ** -18) * 2.0 ** -24 * np.abs(exact)[keep])
    assert got[5] == 1.0 and got[6] == -1.0 and got[7] == 2.0 ** 20
    # arguments the tier must reject still come out as something its range check rejects (zero, infinity or NaN)
    odd = rt.selftest_math(4, np.ones(4), np.array([0.0, np.inf, np.nan, 1e-300]))
    assert all(not (1e-30 <= abs(v) <= 1e15) for v in odd)


def test_bare_division_and_sqrt_of_the_sphere_test_are_correctly_rounded(rt):
    # the walk's sphere test divides by a = d.d with the compiler's division sequence minus its range scaffolding, and takes
    # the square root the same way; inside the ranges it admits, both must give the host's correctly rounded results
    rng = np.random.default_rng(7)
    n = 1 << 22
    x = np.abs(rng.normal(size=n)) * 2.0 ** rng.integers(-299, 299, n)          # a: 2^-300 .. 2^300
    x[: n // 2] = np.abs(rng.normal(size=n // 2)) * 10.0 ** rng.uniform(-3, 3, n // 2) + 1e-9   # where directions live
    num = rng.normal(size=n) * 2.0 ** rng.integers(-300, 299, n)
    num[: n // 2] = rng.normal(size=n // 2) * 10.0 ** rng.uniform(-6, 6, n // 2)
    num[:4] = [0.0, 1.0, -1.0, 2.0 ** -303]
    x[:4] = [3.0, 3.0, 7.0, 2.0 ** 300]
    keep = (x >= 2.0 ** -300) & (x <= 2.0 ** 300) & ((num == 0.0) | ((np.abs(num) >= 2.0 ** -303) & (np.abs(num) <= 2.0 ** 301)))
    got = rt.selftest_math(5, num, x)
    assert np.array_equal(bits(got)[keep], bits(num / x)[keep])
    d = np.abs(rng.normal(size=n)) * 2.0 ** rng.integers(-499, 499, n)
    d[: n // 2] = np.abs(rng.normal(size=n // 2)) * 10.0 ** rng.uniform(-8, 8, n // 2)
    d[:6] = [1.0, 2.0, 4.0, 2.0 ** -500, 2.0 ** 500, 0.25]
    d = np.clip(d, 2.0 ** -500, 2.0 ** 500)
    assert np.array_equal(bits(rt.selftest_math(6, d, d)), bits(np.sqrt(d)))
    # the reciprocal of a root (Vector3::to_unit in a bounce: 1.0 / sqrt(v.v) with the root between 2^-250 and 2^250), and either sign
    root = np.sqrt(d)
    assert np.array_equal(bits(rt.selftest_math(5, np.ones(n), root)), bits(1.0 / root))
    sx = x * rng.choice([-1.0, 1.0], n)
    got = rt.selftest_math(5, num, sx)
    assert np.array_equal(bits(got)[keep], bits(num / sx)[keep])


def test_filtered_quotient_error_bound(rt):
    # the filtered box test relies on |a * RN(1/b) - RN(a/b)| <= 3 * 2^-53 |a/b|
    rng = np.random.default_rng(1)
    n = 1 << 20
    a = rng.normal(size=n) * 10.0 ** rng.integers(-6, 6, n)
    b = rng.normal(size=n) * 10.0 ** rng.integers(-6, 6, n)
    q = rt.selftest_math(2, a, b)
    exact = a / b
    assert np.all(np.abs(q - exact) <= 3 * 2.0 ** -53 * np.abs(exact))
    assert np.any(q != exact)   # it is NOT the same number, which is why the filter keeps a margin


def test_device_rng_stream(rt, oracle):
    for seed, pixel, sample in ((0, 0, 0), (0xC0FFEE, 123456, 499), (2 ** 64 - 1, 2 ** 32 - 2, 7)):
        assert np.array_equal(rt.selftest_rng(seed, pixel, sample, 64), oracle.rng_stream(seed, pixel, sample, 64))


# ------------------------------------------------------------------ K1: primary rays
@pytest.mark.parametrize("name", EXAMPLES)
def test_primary_rays_bit_exact(rt, oracle, name):
    _, cam = S.EXAMPLES[name]()
    cam = dict(cam, image_width=min(cam["image_width"], 480))
    a, b = rt.camera_initialize(cam), oracle.camera_initialize(cam)
    for sample in (0, 17):
        go, gd = rt.primary_rays(a, 99, sample)
        wo, wd = oracle.primary_rays(b, 99, sample)
        assert np.array_equal(bits(go), bits(wo)) and np.array_equal(bits(gd), bits(wd))


# ------------------------------------------------------------------ K2: closest hit
@pytest.mark.parametrize("flags", [0, 16, 8, 4], ids=["f32_lds", "f32_global", "f64_filter", "exact_divide"])
@pytest.mark.parametrize("name", EXAMPLES)
def test_trace_closest_golden_vectors(rt, built, name, flags):
    """Committed vectors: ray batches with the oracle's ids and distances."""
    v = np.load(os.path.join(GOLD, "trace_vectors.npz"))
    _, _, _, dev = built(name)
    ids, t, st = dev.trace_closest(v[name + "_org"], v[name + "_dir"], flags=flags | rt.FLAG_COUNT_WORK)
    assert np.array_equal(ids, v[name + "_id"])
    assert np.array_equal(bits(t), bits(v[name + "_t"]))
    assert [st.node_visits, st.sphere_tests, st.planar_tests] == v[name + "_counts"].tolist()
    assert st.rays == len(ids)


@pytest.mark.parametrize("name", EXAMPLES)
def test_trace_closest_live_oracle(rt, oracle, built, name):
    """Primary rays at the example's own resolution (capped) and two bounces of
    rays leaving the hit points, against the oracle run on the same batch."""
    surfaces, cam, orc, dev = built(name)
    cam = dict(cam, image_width=min(cam["image_width"], 640))
    icam = rt.camera_initialize(cam)
    org, d = rt.primary_rays(icam, 7, 3)
    rng = np.random.default_rng(11)
    for bounce in range(3):
        ids, t, st = dev.trace_closest(org, d)
        wid, wt, ctr = orc.trace_closest(org, d)
        assert np.array_equal(ids, wid), f"bounce {bounce}: {np.count_nonzero(ids != wid)} ids differ"
        assert np.array_equal(bits(t), bits(wt))
        hit = ids != NONE
        if not hit.any():
            break
        org = org[hit] + d[hit] * t[hit, None]
        d = rng.normal(size=org.shape) * rng.choice([1e-3, 1.0, 50.0], size=(len(org), 1))


@pytest.mark.parametrize("batch", ["0", "1"], ids=["f64_boxes_in_loop", "f64_boxes_batched"])
@pytest.mark.parametrize("name", ["cornell_box", "cover_spheres", "quads"])
def test_undecided_boxes_in_loop_and_batched(rt, oracle, name, batch, monkeypatch):
    """The box tests the f32 tier leaves open are resolved inside the walk's loop or for all waiting lanes together
    (chosen per scene, here forced both ways): same hits, same distances, same number of box and surface tests."""
    monkeypatch.setenv("RTX_BATCH_F64_BOXES", batch)
    surfaces, cam = S.EXAMPLES[name]()
    dev, orc = rt.Scene(rt.HostBVH(surfaces)), oracle.Scene(surfaces)
    icam = rt.camera_initialize(dict(cam, image_width=min(cam["image_width"], 480)))
    org, d = rt.primary_rays(icam, 3, 1)
    rng = np.random.default_rng(5)
    for bounce in range(3):
        ids, t, st = dev.trace_closest(org, d, flags=rt.FLAG_COUNT_WORK)
        wid, wt, ctr = orc.trace_closest(org, d)
        assert np.array_equal(ids, wid) and np.array_equal(bits(t), bits(wt)), f"bounce {bounce}"
        assert [st.node_visits, st.sphere_tests, st.planar_tests] == [ctr.node_visits, ctr.sphere_tests, ctr.planar_tests]
        if name == "cornell_box":
            assert st.box_fallbacks > 0.02 * st.node_visits   # this scene does exercise the f64 tiers
        hit = ids != NONE
        if not hit.any():
            break
        org = org[hit] + d[hit] * t[hit, None]
        d = rng.normal(size=org.shape)


@pytest.mark.parametrize("name", ["cover_spheres", "demo_spheres", "simple_scene", "soup_spheres"])
def test_trace_closest_equals_the_brute_force_list(rt, oracle, name):
    """`impl Hittable for &[Surface]` (src/surface.rs:40-59: every surface tested, strict `<`, first wins ties) needs no tree at
    all.  For spheres the tree's boxes enclose their surfaces, so the tree can only skip surfaces that the list would reject
    as well: same hit and same distance wherever no two surfaces tie."""
    if name == "soup_spheres":
        surfaces, cam = S.synthetic_soup(n=3000, kind="spheres", seed=21, image_width=320)
    else:
        surfaces, cam = S.EXAMPLES[name]()
        cam = dict(cam, image_width=min(cam["image_width"], 320))
    dev, orc = rt.Scene(rt.HostBVH(surfaces)), oracle.Scene(surfaces)
    org, d = rt.primary_rays(rt.camera_initialize(cam), 31, 0)
    rng = np.random.default_rng(17)
    for bounce in range(2):
        ids, t, _ = dev.trace_closest(org, d)
        lid, lt = orc.trace_bruteforce(org, d)
        same_t = bits(t) == bits(np.where(lid == NONE, np.inf, lt))
        assert same_t.all()
        assert (ids == lid)[same_t].mean() > 0.9999      # equal distances to two different spheres: the two orders may differ
        hit = ids != NONE
        org = org[hit] + d[hit] * t[hit, None]
        d = rng.normal(size=org.shape)


def test_trace_closest_interval_and_ties(rt, oracle):
    lam = S.lambertian((1, 1, 1))
    surf = S.pack([S.quad((0, 0, 0), (1, 0, 0), (0, 1, 0), lam), S.quad((0, 0, 0), (1, 0, 0), (0, 1, 0), lam),
                   S.quad((5, 0, 0), (1, 0, 0), (0, 1, 0), lam), S.sphere((0.5, 0.5, -3), 1.0, lam), S.sphere((0.5, 0.5, -3), 1.0, lam)])
    orc = oracle.Scene(surf, oracle.BISECT)
    dev = rt.Scene(rt.HostBVH(surf, rt.BISECT))
    org = np.array([[0.5, 0.5, 1.0], [0.5, 0.5, 1.0], [0.5, 0.5, -9.0], [5.5, 0.5, 1.0], [0.0, 0.0, 1.0], [9, 9, 9]])
    d = np.array([[0, 0, -1.0], [0, 0, -2.0], [0, 0, 1.0], [0, 0, -1.0], [0, 0, -1.0], [0, 0, 1.0]])
    for tmin, tmax in ((0.001, np.inf), (0.001, 1.0), (1.0, 1.0), (1.0000001, 10.0), (0.001, 0.4)):
        ids, t, _ = dev.trace_closest(org, d, tmin, tmax)
        wid, wt, _ = orc.trace_closest(org, d, tmin, tmax)
        assert np.array_equal(ids, wid) and np.array_equal(bits(t), bits(np.where(wid == NONE, np.inf, wt)))


def test_trace_closest_degenerate_rays(rt, oracle, built):
    """Zero direction components, rays inside boxes, zero and non-finite
    directions: the filtered test must agree with the division form everywhere."""
    for name in ("cornell_box", "cover_spheres", "quads"):
        surfaces, cam, orc, dev = built(name)
        rng = np.random.default_rng(5)
        n = 20000
        lo, hi = (0, 555) if name == "cornell_box" else (-12, 12)
        org = rng.uniform(lo, hi, (n, 3))
        d = rng.normal(size=(n, 3))
        d[rng.random((n, 3)) < 0.3] = 0.0                    # axis-parallel and zero directions
        org[: n // 4] = np.round(org[: n // 4])              # origins on integer planes (box faces in cornell)
        d[-8:] = [[np.inf, 1, 1], [np.nan, 0, 1], [1e-320, 1, 0], [1e305, 1, 1], [0, 0, 0], [-0.0, 1, 0], [1, -np.inf, 0], [1e-310, 1e-310, 1e-310]]
        wid, wt, _ = orc.trace_closest(org, d)
        for flags in (0, rt.FLAG_NO_LDS, rt.FLAG_F64_FILTER, rt.FLAG_EXACT_DIVIDE):
            ids, t, _ = dev.trace_closest(org, d, flags=flags)
            assert np.array_equal(ids, wid), flags
            assert np.array_equal(bits(t), bits(np.where(wid == NONE, np.inf, wt))), flags


def test_empty_and_single_surface_scene(rt, oracle):
    dev = rt.Scene(rt.HostBVH(S.pack([])))
    ids, t, _ = dev.trace_closest([[0, 0, 0]], [[0, 0, -1]])
    assert ids[0] == NONE and t[0] == np.inf
    cam = rt.camera_initialize(S.camera(image_width=16, samples_per_pixel=2, background=(0.25, 0.5, 1.0)))
    img, st = dev.render(cam, 1)
    assert np.array_equal(img, np.broadcast_to([0.25, 0.5, 1.0], img.shape)) and st.rays == 16 * 16 * 2
    one = S.pack([S.sphere((0, 0, -5), 1, S.lambertian((1, 1, 1)))])
    dev = rt.Scene(rt.HostBVH(one))
    ids, t, _ = dev.trace_closest([[0, 0, 0], [0, 3, 0]], [[0, 0, -1], [0, 0, -1]])
    assert ids.tolist() == [0, NONE] and t[0] == 4.0


# ------------------------------------------------------------------ K1-K5: the render loop
def render_pair(rt, oracle, surfaces, cam, seed, spp_begin, spp_count, iterative, strategy=3):
    import torch

    dev = rt.Scene(rt.HostBVH(surfaces, strategy))
    orc = oracle.Scene(surfaces, strategy)
    icam, ocam = rt.camera_initialize(cam), oracle.camera_initialize(cam)
    acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    st = dev.render_accumulate(icam, seed, spp_begin, spp_count, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    want, ctr = orc.render(ocam, seed, spp_begin, spp_count, iterative=iterative)
    return acc.cpu().numpy(), st, want, ctr


@pytest.mark.parametrize("name", EXAMPLES)
def test_single_sample_radiance_bit_exact(rt, oracle, name):
    """One sample per pixel: the accumulation buffer holds exactly one path's
    radiance per pixel, so the whole pipeline (ray generation, traversal,
    hit completion, emitted, scatter, RNG draw order, depth limit) is compared
    bit for bit with the oracle's loop form of ray_color."""
    surfaces, cam = S.EXAMPLES[name]()
    cam = dict(cam, image_width=min(cam["image_width"], 320))
    for seed, sample in ((1, 0), (0xC0FFEE, 41)):
        got, st, want, ctr = render_pair(rt, oracle, surfaces, cam, seed, sample, 1, iterative=True)
        assert st.rays == ctr.rays and st.paths == ctr.paths
        assert np.array_equal(bits(got + 0.0), bits(want + 0.0)), f"{np.count_nonzero(got != want)} values differ"


@pytest.mark.parametrize("name", EXAMPLES)
def test_single_sample_radiance_bit_exact_with_the_paths_in_registers(rt, oracle, name, monkeypatch):
    """The same comparison with the resident form of the single launch forced on every example scene (a tree of one node,
    scenes of quads whose undecided box tests are batched, small sphere scenes): both instances of that kernel, and its
    node loop, against the oracle bit for bit."""
    monkeypatch.setenv("RTX_RESIDENT", "1")
    surfaces, cam = S.EXAMPLES[name]()
    cam = dict(cam, image_width=min(cam["image_width"], 240))
    for seed, sample in ((7, 0), (0xBEEF, 13)):
        got, st, want, ctr = render_pair(rt, oracle, surfaces, cam, seed, sample, 1, iterative=True)
        assert st.kernel_form == rt.FORM_RESIDENT
        assert st.rays == ctr.rays and st.paths == ctr.paths
        assert np.array_equal(bits(got + 0.0), bits(want + 0.0)), f"{np.count_nonzero(got != want)} values differ"


@pytest.mark.parametrize("path", ["single_launch", "round_loop"])
@pytest.mark.parametrize("name", ["cover_spheres", "cornell_box", "demo_spheres"])
def test_render_work_counters_match_oracle(rt, oracle, name, path, monkeypatch):
    """Both render paths (one persistent launch; generate/traverse/shade rounds) do exactly the
    oracle's box and surface tests, and produce the same single-sample image."""
    import torch

    monkeypatch.setenv("RTX_PERSISTENT", "1" if path == "single_launch" else "0")
    surfaces, cam = S.EXAMPLES[name]()
    cam = dict(cam, image_width=min(cam["image_width"], 256))
    dev = rt.Scene(rt.HostBVH(surfaces))
    orc = oracle.Scene(surfaces)
    icam, ocam = rt.camera_initialize(cam), oracle.camera_initialize(cam)
    acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    st = dev.render_accumulate(icam, 5, 3, 1, acc.data_ptr(), torch.cuda.current_stream().cuda_stream, rt.FLAG_COUNT_WORK)
    want, ctr = orc.render(ocam, 5, 3, 1, iterative=True)
    assert [st.rays, st.paths, st.node_visits, st.sphere_tests, st.planar_tests] == \
           [ctr.rays, ctr.paths, ctr.node_visits, ctr.sphere_tests, ctr.planar_tests]
    assert np.array_equal(bits(acc.cpu().numpy() + 0.0), bits(want + 0.0))


@pytest.mark.parametrize("name", ["cover_spheres", "cornell_box", "demo_spheres", "simple_light"])
def test_radiance_matches_recursive_ray_color(rt, oracle, name):
    """Against ray_color exactly as the reference writes it (recursive): only
    the association of the attenuation product differs."""
    surfaces, cam = S.EXAMPLES[name]()
    cam = dict(cam, image_width=200)
    got, st, want, ctr = render_pair(rt, oracle, surfaces, cam, 5, 2, 6, iterative=False)
    assert st.rays == ctr.rays
    assert np.allclose(got, want, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("n,kind,wide", [(1500, "spheres", "1"), (2300, "spheres", "1"), (900, "triangles", "1"), (1500, "spheres", "0")])
def test_mid_size_scenes_in_the_one_block_per_cu_form(rt, oracle, n, kind, wide, monkeypatch):
    """Trees too large for five LDS copies per CU but small enough for one (1500 spheres: tree and spheres in LDS;
    2300 spheres and 900 triangles: the tree alone), and the same scene from global memory: same paths, bit for bit."""
    import torch

    monkeypatch.setenv("RTX_LDS_WIDE_BYTES", str(150 * 1024) if wide == "1" else "0")
    surfaces, cam = S.synthetic_soup(n=n, kind=kind, seed=7, image_width=192, samples_per_pixel=1)
    cam = dict(cam, max_depth=6)
    dev = rt.Scene(rt.HostBVH(surfaces))
    orc = oracle.Scene(surfaces)
    icam, ocam = rt.camera_initialize(cam), oracle.camera_initialize(cam)
    acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    st = dev.render_accumulate(icam, 11, 0, 1, acc.data_ptr(), torch.cuda.current_stream().cuda_stream, rt.FLAG_COUNT_WORK)
    want, ctr = orc.render(ocam, 11, 0, 1, iterative=True)
    assert [st.rays, st.paths, st.node_visits, st.sphere_tests, st.planar_tests] == \
           [ctr.rays, ctr.paths, ctr.node_visits, ctr.sphere_tests, ctr.planar_tests]
    assert np.array_equal(bits(acc.cpu().numpy() + 0.0), bits(want + 0.0))


@pytest.mark.parametrize("form", ["resident", "pooled"])
def test_paths_resident_in_registers_render_the_same_image(rt, oracle, form, monkeypatch):
    """The render kernel that keeps every path in registers from bounce to bounce (scenes that sit in LDS whole) against the
    oracle and against the pooled kernel: one sample per pixel bit for bit (with the rays counted), several samples up to the
    order of the f64 sums, depth limits from 1 up, a frame with fewer paths than one wave has lanes, and a soup of 600 spheres."""
    monkeypatch.setenv("RTX_RESIDENT", "1" if form == "resident" else "0")
    surfaces, cam = S.cover_spheres()
    for width, seed, sample in ((320, 1, 0), (200, 0xC0FFEE, 41), (7, 5, 3)):
        got, st, want, ctr = render_pair(rt, oracle, surfaces, dict(cam, image_width=width), seed, sample, 1, iterative=True)
        assert st.rays == ctr.rays and st.paths == ctr.paths
        assert np.array_equal(bits(got + 0.0), bits(want + 0.0)), f"{np.count_nonzero(got != want)} values differ"
    for max_depth in (1, 2, 7, 50):
        got, st, want, ctr = render_pair(rt, oracle, surfaces, dict(cam, image_width=96, max_depth=max_depth), 3, 2, 4, iterative=True)
        assert st.rays == ctr.rays
        assert np.allclose(got, want, rtol=1e-13, atol=0)
    soup, scam = S.synthetic_soup(n=600, kind="spheres", seed=7, image_width=192, samples_per_pixel=1)
    got, st, want, ctr = render_pair(rt, oracle, soup, dict(scam, max_depth=6), 11, 0, 1, iterative=True)
    assert st.rays == ctr.rays
    assert np.array_equal(bits(got + 0.0), bits(want + 0.0))


@pytest.mark.parametrize("form", ["resident", "pooled"])
def test_render_far_from_the_origin_where_the_first_tier_often_cannot_decide(rt, oracle, form, monkeypatch):
    """A field of small spheres 3e5 away from the origin: single-precision box planes are then uncertain by hundredths of a
    unit, many box tests are left to the exact tiers (counted), and a lane that stood on such a box has to come back to it.
    Both forms of the single launch give the oracle's samples bit for bit and trace the oracle's number of rays."""
    import torch

    shift = np.array([3.0e5, -2.0e5, 2.5e5])
    soup, cam = S.synthetic_soup(n=700, kind="spheres", seed=19, image_width=160, samples_per_pixel=1)
    soup = soup.copy()
    soup["q"] = soup["q"] * 0.08 + shift          # 700 spheres of radius 0.05 .. 0.3 in a cube of side 8
    cam = dict(cam, look_from=tuple(shift + np.array([0.0, 0.0, -14.0])), look_at=tuple(shift), max_depth=8)
    monkeypatch.setenv("RTX_RESIDENT", "1" if form == "resident" else "0")
    for seed, sample in ((4, 0), (99, 7)):
        got, st, want, ctr = render_pair(rt, oracle, soup, cam, seed, sample, 1, iterative=True)
        assert st.rays == ctr.rays and st.paths == ctr.paths and st.rays > 1.2 * st.paths
        assert st.kernel_form == (rt.FORM_RESIDENT if form == "resident" else rt.FORM_POOLED)
        assert np.array_equal(bits(got + 0.0), bits(want + 0.0)), f"{np.count_nonzero(got != want)} values differ"
    # the scene does what it is for: a good share of the walk's box tests go to the exact tiers
    dev = rt.Scene(rt.HostBVH(soup))
    icam = rt.camera_initialize(cam)
    acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    counted = dev.render_accumulate(icam, 4, 0, 1, acc.data_ptr(), torch.cuda.current_stream().cuda_stream, rt.FLAG_COUNT_ACTUAL)
    assert counted.box_fallbacks > 0.02 * counted.node_visits


def test_first_call_of_a_scene_measures_its_rays_per_path_with_its_first_sample(rt, oracle, monkeypatch):
    """A scene's first render call of eight samples or more runs its first sample as a launch of its own (which tells how many
    rays a path of the scene traces) and the others in the form that suits: the image and the ray count are those of the oracle
    and of the same call with that turned off, the device time covers both launches, and a second call gives the same image."""
    import torch

    surfaces, cam = S.cover_spheres()
    cam = dict(cam, image_width=160)
    icam = rt.camera_initialize(cam)
    stream = torch.cuda.current_stream().cuda_stream
    images, stats = [], []
    for probe in ("1", "0", "1"):
        monkeypatch.setenv("RTX_PROBE_FIRST_CALL", probe)
        dev = rt.Scene(rt.HostBVH(surfaces))
        acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
        stats.append(dev.render_accumulate(icam, 9, 5, 12, acc.data_ptr(), stream))
        images.append(acc.cpu().numpy())
        if probe == "1":
            again = torch.zeros_like(acc)
            st2 = dev.render_accumulate(icam, 9, 5, 12, again.data_ptr(), stream)
            assert st2.rays == stats[-1].rays
            assert np.allclose(again.cpu().numpy(), images[-1], rtol=1e-13, atol=0)
        dev.close()
    want, ctr = oracle.Scene(surfaces).render(oracle.camera_initialize(cam), 9, 5, 12, iterative=True)
    for img, st in zip(images, stats):
        assert st.rays == ctr.rays and st.paths == ctr.paths == 160 * 90 * 12
        assert st.ms_total > 0
        assert np.allclose(img, want, rtol=1e-13, atol=0)
    # fewer than eight samples: one launch, as before
    dev = rt.Scene(rt.HostBVH(surfaces))
    acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    st = dev.render_accumulate(icam, 9, 5, 7, acc.data_ptr(), stream)
    want7, ctr7 = oracle.Scene(surfaces).render(oracle.camera_initialize(cam), 9, 5, 7, iterative=True)
    assert st.rays == ctr7.rays
    assert np.allclose(acc.cpu().numpy(), want7, rtol=1e-13, atol=0)


def test_depth_limit_and_pool_smaller_than_job(rt, oracle, monkeypatch):
    surfaces, cam = S.cornell_box()
    for max_depth in (1, 2, 7):
        c = dict(cam, image_width=96, max_depth=max_depth)
        got, st, want, ctr = render_pair(rt, oracle, surfaces, c, 3, 0, 3, iterative=True)
        assert st.rays == ctr.rays
        assert np.allclose(got, want, rtol=1e-13, atol=0)
    # a pool far smaller than the job: slots are restarted on new (pixel, sample) pairs many times over
    monkeypatch.setenv("RTX_POOL_SLOTS", "4096")
    surfaces, cam = S.cover_spheres()
    c = dict(cam, image_width=128)
    got, st, want, ctr = render_pair(rt, oracle, surfaces, c, 8, 10, 5, iterative=True)
    assert st.rays == ctr.rays and st.paths == 128 * 72 * 5
    assert np.allclose(got, want, rtol=1e-13, atol=0)


def test_all_partition_strategies_render_the_same_paths(rt, oracle):
    surfaces, cam = S.simple_scene()
    cam = dict(cam, image_width=120)
    for strategy in (rt.BISECT, rt.MIDPOINT, rt.SAH_PER_SURFACE):
        got, st, want, ctr = render_pair(rt, oracle, surfaces, cam, 2, 0, 1, iterative=True, strategy=strategy)
        assert np.array_equal(bits(got + 0.0), bits(want + 0.0))


def test_sample_ranges_add_up(rt):
    """Splitting the sample range over calls (what the multi-GPU run does over
    ranks) gives the image of the unsplit call up to f64 summation order."""
    import torch

    surfaces, cam = S.cover_spheres()
    icam = rt.camera_initialize(dict(cam, image_width=256))
    dev = rt.Scene(rt.HostBVH(surfaces))
    stream = torch.cuda.current_stream().cuda_stream
    whole = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    parts = torch.zeros_like(whole)
    s0 = dev.render_accumulate(icam, 4, 0, 8, whole.data_ptr(), stream)
    rays = 0
    for b, c in ((0, 3), (3, 3), (6, 2)):
        rays += dev.render_accumulate(icam, 4, b, c, parts.data_ptr(), stream).rays
    assert rays == s0.rays
    assert torch.allclose(whole, parts, rtol=1e-13, atol=0)
    again = torch.zeros_like(whole)
    dev.render_accumulate(icam, 4, 0, 8, again.data_ptr(), stream)
    assert torch.allclose(whole, again, rtol=1e-13, atol=0)
    other_seed = torch.zeros_like(whole)
    dev.render_accumulate(icam, 5, 0, 8, other_seed.data_ptr(), stream)
    assert not torch.allclose(whole, other_seed, rtol=1e-3)


def test_render_accumulate_without_waiting(rt):
    """RTX_FLAG_NO_WAIT: the call returns once its launch is enqueued; ten calls in a row on one stream (more than the
    ring of upload records holds) add up to the image of the waiting calls."""
    import torch

    surfaces, cam = S.cover_spheres()
    icam = rt.camera_initialize(dict(cam, image_width=256))
    dev = rt.Scene(rt.HostBVH(surfaces))
    side = torch.cuda.Stream()
    want = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
    got = torch.zeros_like(want)
    torch.cuda.synchronize()
    rays = 0
    for k in range(10):
        rays += dev.render_accumulate(icam, 9, 2 * k, 2, want.data_ptr(), torch.cuda.current_stream().cuda_stream).rays
    with torch.cuda.stream(side):
        for k in range(10):
            st = dev.render_accumulate(icam, 9, 2 * k, 2, got.data_ptr(), side.cuda_stream, rt.FLAG_NO_WAIT)
            assert st.rays == 0 and st.ms_total == 0.0      # nothing is known yet when the call returns
        side.synchronize()
    assert rays > 0 and torch.allclose(got, want, rtol=1e-13, atol=0)
    for bad in (rt.FLAG_COUNT_WORK, rt.FLAG_EXACT_DIVIDE, rt.FLAG_F64_FILTER, rt.FLAG_TIME_KERNELS):
        with pytest.raises(rt.RtxError) as e:
            dev.render_accumulate(icam, 9, 0, 1, got.data_ptr(), side.cuda_stream, rt.FLAG_NO_WAIT | bad)
        assert e.value.code == -1


def test_headline_size_properties(rt, oracle):
    """cover_spheres at the benchmark's full 1920x1080: properties that hold at
    any size (the oracle is too slow to render this size in a unit test)."""
    surfaces, cam = S.cover_spheres(image_width=1920)
    icam = rt.camera_initialize(cam)
    assert (icam.image_width, icam.image_height) == (1920, 1080)
    dev = rt.Scene(rt.HostBVH(surfaces))
    img, st = dev.render(rt.camera_initialize(dict(cam, samples_per_pixel=2)), 0xC0FFEE)
    assert st.paths == 2 * 1920 * 1080 and st.paths < st.rays <= 50 * st.paths
    assert np.isfinite(img).all() and img.min() >= 0.0 and img.max() <= 1.0 + 1e-12   # albedos <= 1, background <= 1
    # first-hit ids of the full-resolution primary batch against the oracle
    org, d = rt.primary_rays(icam, 0xC0FFEE, 0)
    ids, t, _ = dev.trace_closest(org, d)
    wid, wt, _ = oracle.Scene(surfaces).trace_closest(org[::7], d[::7])
    assert np.array_equal(ids[::7], wid) and np.array_equal(bits(t[::7]), bits(np.where(wid == NONE, np.inf, wt)))
    # rows of sky above the horizon are exactly the background colour
    assert np.array_equal(img[0, 0], [0.7, 0.8, 1.0])


def test_rows_without_survivors_and_without_radiance(rt, oracle, monkeypatch):
    """The pooled kernel's one-block-per-CU form issues a row's atomics and stores by inline assembly with a fixed count per
    row and waits by that count; lanes with nothing to add or store are masked out.  With a black sky and a depth limit of one
    EVERY row has an empty mask in all sixteen operations (no path survives its first bounce, nothing is emitted, the sky adds
    zero); with a limit of two the rows of the second bounce do.  Image and ray count are the oracle's and the round loop's,
    whose shade kernel is plain C++."""
    import torch

    surfaces, cam = S.cover_spheres()
    cam = dict(cam, image_width=256, background=(0.0, 0.0, 0.0))
    stream = torch.cuda.current_stream().cuda_stream
    for max_depth in (1, 2):
        c = dict(cam, max_depth=max_depth)
        icam = rt.camera_initialize(c)
        want, ctr = oracle.Scene(surfaces).render(oracle.camera_initialize(c), 6, 2, 3, iterative=True)
        for env in ({"RTX_RESIDENT": "0"}, {"RTX_PERSISTENT": "0"}):
            for k in ("RTX_RESIDENT", "RTX_PERSISTENT"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            dev = rt.Scene(rt.HostBVH(surfaces))
            acc = torch.zeros(icam.image_height, icam.image_width, 3, dtype=torch.float64, device="cuda")
            st = dev.render_accumulate(icam, 6, 2, 3, acc.data_ptr(), stream)
            assert st.kernel_form == (rt.FORM_POOLED if "RTX_RESIDENT" in env else rt.FORM_ROUND_LOOP)
            assert st.rays == ctr.rays and st.paths == ctr.paths
            got = acc.cpu().numpy()
            assert np.allclose(got, want, rtol=1e-13, atol=0)
            if max_depth == 1:
                assert st.rays == st.paths and not got.any()


def test_headline_frame_in_both_forms_of_the_kernel(rt, oracle, monkeypatch):
    """cover_spheres at the benchmark's 1920x1080 in the form the benchmark times (paths resident in registers) and in the
    pooled form: the same rays, paths and image (up to the order of the f64 sums) at 9 samples per pixel -- 1.9e7 paths, so
    the resident form's chunks of path indices run through their whole range of sizes, down to 64 at the end of the call --
    the same again for a fresh scene whose first call goes through the first-sample probe, and the whole frame at one sample
    per pixel against the oracle bit for bit."""
    import torch

    surfaces, cam = S.cover_spheres(image_width=1920)
    icam = rt.camera_initialize(cam)
    assert (icam.image_width, icam.image_height) == (1920, 1080)
    stream = torch.cuda.current_stream().cuda_stream
    bvh = rt.HostBVH(surfaces)

    def frame(resident, begin, count):
        if resident is None:
            monkeypatch.delenv("RTX_RESIDENT", raising=False)
        else:
            monkeypatch.setenv("RTX_RESIDENT", resident)
        dev = rt.Scene(bvh)
        acc = torch.zeros(1080, 1920, 3, dtype=torch.float64, device="cuda")
        st = dev.render_accumulate(icam, 0xC0FFEE, begin, count, acc.data_ptr(), stream)
        dev.close()
        return acc, st

    res, st_res = frame("1", 3, 9)
    pool, st_pool = frame("0", 3, 9)
    probed, st_probed = frame(None, 3, 9)   # a scene's first call: sample 3 as a pooled launch of its own, samples 4..11 resident
    assert st_res.paths == st_pool.paths == st_probed.paths == 9 * 1920 * 1080
    assert st_res.rays == st_pool.rays == st_probed.rays and st_res.rays > 2 * st_res.paths
    assert torch.allclose(res, pool, rtol=1e-13, atol=0) and torch.allclose(res, probed, rtol=1e-13, atol=0)
    assert int((res.sum(dim=2) == 0).sum()) == 0   # every pixel of this scene receives light: none was skipped
    one, st_one = frame("1", 41, 1)
    want, ctr = oracle.Scene(surfaces).render(oracle.camera_initialize(cam), 0xC0FFEE, 41, 1, iterative=True)
    assert st_one.rays == ctr.rays and st_one.paths == ctr.paths == 1920 * 1080
    assert np.array_equal(bits(one.cpu().numpy() + 0.0), bits(want + 0.0))


# ------------------------------------------------------------------ converged images
def test_hello_triangle_golden_png(rt):
    from PIL import Image

    png = np.asarray(Image.open(os.path.join(GOLD, "hello_triangle_10k.png")).convert("RGB")).astype(int)
    surfaces, cam = S.hello_triangle()
    img = rt.run(surfaces, dict(cam, samples_per_pixel=2000), seed=1)
    import torch

    lin = torch.from_numpy(img).cuda()
    q = torch.empty(lin.shape, dtype=torch.uint8, device="cuda")
    rt.quantize(lin.data_ptr(), lin.numel(), q.data_ptr())
    torch.cuda.synchronize()
    diff = np.abs(q.cpu().numpy().astype(int) - png)
    rmse = np.sqrt((diff.astype(float) ** 2).mean())
    assert rmse <= 1.0, rmse            # 8-bit units, both images converged
    assert (diff <= 1).mean() > 0.9995


def test_cornell_box_golden_png(rt, oracle):
    """600x600 Cornell box against the reference's 20,000-spp render.  The
    tolerance is measured, not guessed: RMSE between two of our own renders
    with different seeds at the same spp is the noise floor of one render; the
    golden (far more samples) must lie within 1.25x of it."""
    from PIL import Image

    png = np.asarray(Image.open(os.path.join(GOLD, "cornell_box_20k.png")).convert("RGB")).astype(float)
    surfaces, cam = S.cornell_box(samples_per_pixel=1500)
    a = oracle.ppm_quantize(rt.run(surfaces, cam, seed=1)).astype(float)
    b = oracle.ppm_quantize(rt.run(surfaces, cam, seed=2)).astype(float)
    floor = np.sqrt(((a - b) ** 2).mean()) / np.sqrt(2.0)   # noise of ONE render
    rmse = np.sqrt(((a - png) ** 2).mean())
    print(f"cornell golden: rmse {rmse:.4f}, noise floor of one render {floor:.4f}, ratio {rmse / floor:.4f}")
    assert rmse <= 1.25 * floor, (rmse, floor)
    assert abs(a.mean() - png.mean()) < 1.0                  # no brightness bias


def test_quantize_matches_oracle(rt, oracle):
    import torch

    x = np.concatenate([np.linspace(-0.5, 1.5, 4001), [np.nan, np.inf, -np.inf, 0.0, 1e-300, 0.998001, 0.998002]])
    x = np.resize(x, (x.size // 3) * 3)
    d = torch.from_numpy(x).cuda()
    q = torch.empty(d.shape, dtype=torch.uint8, device="cuda")
    rt.quantize(d.data_ptr(), d.numel(), q.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), oracle.ppm_quantize(x.reshape(-1, 3)).reshape(-1))
    out = torch.empty_like(d)
    rt.scale(d.data_ptr(), d.numel(), 0.002, out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(bits(out.cpu().numpy()), bits(x * 0.002))


def expected_ppm(oracle, img):
    """The reference's output (src/camera.rs:129-131,155-161) from the oracle's ppm_pixel values."""
    h, w, _ = img.shape
    q = oracle.ppm_quantize(img.reshape(-1, 3))
    lines = "".join(f"{r} {g} {b}\n" for r, g, b in q.tolist())
    return f"P3\n{w} {h}\n255\n{lines}".encode()


@pytest.mark.parametrize("shape", [(1, 1), (3, 17), (64, 16), (135, 240), (1080, 1920)])
def test_ppm_text_is_the_references_output(rt, oracle, shape):
    import torch

    rng = np.random.default_rng(shape[0] * 7919 + shape[1])
    img = rng.uniform(-0.2, 1.3, shape + (3,)) ** 2   # values on both sides of every digit-count boundary and of the clamp
    special = [np.nan, np.inf, -np.inf, 0.0, 1e-300, 0.998001, 0.998002]
    k = min(img.size, len(special))
    img.reshape(-1)[:k] = special[:k]
    d = torch.from_numpy(img).cuda()
    got = rt.ppm_text(d.data_ptr(), shape[1], shape[0])
    assert got == expected_ppm(oracle, img)


def test_runner_writes_the_references_ppm(rt, oracle, tmp_path):
    """RenderRunner::run (rth_run): the file is the P3 text of the image it returns."""
    import ctypes as C

    surfaces, cam = S.EXAMPLES["quads"]()
    cfg = rt.camera_config(dict(cam, samples_per_pixel=4, image_width=160))
    icam = rt.camera_initialize(cfg)
    img = np.zeros((icam.image_height, icam.image_width, 3))
    path = tmp_path / "out.ppm"
    rc = rt._lib.rth_run(surfaces.ctypes.data_as(C.c_void_p), len(surfaces), 3, 0, C.byref(cfg), 5, str(path).encode(),
                         img.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, rt._lib.rth_last_error()
    assert path.read_bytes() == expected_ppm(oracle, img)


def test_example_program_prints_the_runners_ppm(rt, oracle, tmp_path):
    """bin/rtx_example (the reference's `cargo run --example`): stdout is the P3 text of the image
    rth_run returns for the same scene, camera and seed."""
    import ctypes as C
    import subprocess

    if not os.path.exists(os.path.join(ROOT, "bin", "rtx_example")):
        pytest.skip("bin/rtx_example is not built (run make)")
    surfaces, cfg = rt.example("simple_scene")
    cfg.image_width, cfg.samples_per_pixel = 96, 3
    icam = rt.camera_initialize(cfg)
    img = np.zeros((icam.image_height, icam.image_width, 3))
    assert rt._lib.rth_run(surfaces.ctypes.data_as(C.c_void_p), len(surfaces), 3, 0, C.byref(cfg), 9, None,
                           img.ctypes.data_as(C.POINTER(C.c_double))) == 0
    for build in ("host", "device"):
        p = subprocess.run([os.path.join(ROOT, "bin", "rtx_example"), "simple_scene", "--width", "96", "--spp", "3", "--seed", "9", "--build", build],
                           capture_output=True, timeout=120)
        assert p.returncode == 0, p.stderr.decode()
        assert b"BVH construction" in p.stderr and b"Rendering" in p.stderr
        got = np.array(p.stdout.split()[4:], dtype=np.int64).reshape(img.shape)
        want = oracle.ppm_quantize(img.reshape(-1, 3)).reshape(img.shape)
        assert p.stdout.startswith(b"P3\n96 54\n255\n")
        assert np.abs(got - want).max() <= 1   # the two runs add the same radiance values in different orders
