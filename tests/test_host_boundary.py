"""CPU tests of the product's host side and of the C-ABI library itself:
the host BVH builder and camera set-up against the oracle, and that
librtx.so loads and exports every symbol include/*.h declares.  No compute
entry point is called here (there is no GPU in this environment)."""
import ctypes
import os
import re

import numpy as np
import pytest

from raytracing_amd import scenes as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"\b((?:rtx|rth)_[a-z0-9_]+)\s*\(", text)


def test_library_exports_every_declared_symbol(rt):
    lib = ctypes.CDLL(rt.LIB_PATH)
    declared = set(declared_functions("rtx.h")) | set(declared_functions("rtx_host.h"))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported by librtx.so"
    assert declared == set(rt.ABI_SYMBOLS)


def test_struct_layouts_match_headers(rt, oracle):
    assert ctypes.sizeof(rt.Camera) == 16 + 8 + 24 + 8 + 6 * 24
    assert ctypes.sizeof(rt.CameraConfig) == ctypes.sizeof(oracle.CameraConfig) == 8 + 16 + 8 + 72 + 16 + 24
    assert ctypes.sizeof(rt.Stats) == 7 * 8 + 4 * 8 + 2 * 4
    assert ctypes.sizeof(rt.SceneDesc) == 16 + 17 * 8
    assert ctypes.sizeof(rt.MultiStats) == 16 + 8 + 16 + rt.MULTI_MAX_GPUS * (4 + 4 + 8 + 8)
    assert rt.SURFACE_DTYPE.itemsize == 120


STRATEGIES = [("bisect", 0, 0), ("midpoint", 1, 0), ("sah_equal_16", 2, 16), ("sah_per_surface", 3, 0)]


@pytest.mark.parametrize("name", list(S.EXAMPLES))
@pytest.mark.parametrize("strategy", STRATEGIES, ids=[s[0] for s in STRATEGIES])
def test_host_builder_equals_oracle_builder(rt, oracle, name, strategy):
    surfaces, _ = S.EXAMPLES[name]()
    _, strat, buckets = strategy
    try:
        want = oracle.Scene(surfaces, strat, buckets).node_tables()
    except oracle.OracleError as e:
        with pytest.raises(rt.RtxError) as ei:
            rt.HostBVH(surfaces, strat, buckets)
        key = "No valid splitting plane" if "No valid" in str(e) else "side empty"
        assert ei.value.code == -5 and key in str(ei.value)
        return
    got = rt.HostBVH(surfaces, strat, buckets).node_tables()
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and np.array_equal(g, w)
    n = len(surfaces)
    assert len(got[2]) == 2 * n - 1


def test_host_builder_random_scenes(rt, oracle):
    rng = np.random.default_rng(3)
    for trial in range(6):
        n = int(rng.integers(2, 400))
        surf, _ = S.synthetic_soup(n, "triangles" if trial % 2 else "spheres", seed=trial)
        for strat, buckets in ((1, 0), (2, 8), (3, 0)):
            try:
                want = oracle.Scene(surf, strat, buckets).node_tables()
            except oracle.OracleError:
                with pytest.raises(rt.RtxError):
                    rt.HostBVH(surf, strat, buckets)
                continue
            got = rt.HostBVH(surf, strat, buckets).node_tables()
            assert all(np.array_equal(g, w) for g, w in zip(got, want))


def test_host_builder_failures(rt):
    lam = S.lambertian((1, 1, 1))
    with pytest.raises(rt.RtxError, match="No valid splitting plane") as e:
        rt.HostBVH(S.pack([S.sphere((0, 0, 0), 1, lam), S.sphere((0, 0, 0), 2, lam)]))
    assert e.value.code == -5
    with pytest.raises(rt.RtxError, match="invalid radius"):
        rt.HostBVH(S.pack([S.sphere((0, 0, 0), -1, lam)]))
    empty = rt.HostBVH(S.pack([]))
    assert empty.n_nodes == 0
    single = rt.HostBVH(S.pack([S.sphere((0, 0, 0), 1, lam)]))
    assert single.n_nodes == 1


@pytest.mark.parametrize("name", list(S.EXAMPLES))
def test_camera_initialize_equals_oracle(rt, oracle, name):
    _, cam = S.EXAMPLES[name]()
    for width in (cam["image_width"], 1920, 37, 1):
        c = dict(cam, image_width=width)
        a, b = rt.camera_initialize(c), oracle.camera_initialize(c)
        assert bytes(a) == bytes(b)


def test_cover_spheres_fixture(rt):
    # the benchmark scene is a fixed function of its seed
    surfaces, cam = S.cover_spheres()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "cover_spheres_seed_c0ffee.npz"))["surfaces"]
    assert surfaces.tobytes() == fx.tobytes()
    assert 440 < len(surfaces) <= 488
    kinds = np.bincount(surfaces["mat_kind"], minlength=3)
    assert kinds[0] > kinds[1] > kinds[2] > 0
    assert np.array_equal(surfaces[0]["q"], [0, -1000, 0]) and surfaces[-1]["mat_kind"] == S.METAL


def test_render_api_rejects_bad_arguments_without_a_device(rt):
    # argument checks come before any device work, so they run anywhere
    lib = ctypes.CDLL(rt.LIB_PATH)
    lib.rtx_last_error.restype = ctypes.c_char_p
    assert lib.rtx_scene_create(None, None) == -1
    assert b"null" in lib.rtx_last_error()
    assert lib.rtx_trace_closest(None, None, None, ctypes.c_uint64(0), ctypes.c_double(0), ctypes.c_double(1), 0, None, None, None) == -1
    bad = rt.SceneDesc()
    bad.n_nodes = 1
    out = ctypes.c_void_p()
    assert lib.rtx_scene_create(ctypes.byref(bad), ctypes.byref(out)) == -1


@pytest.mark.parametrize("name", list(S.EXAMPLES))
def test_cpp_example_scenes_equal_the_python_ones(rt, name):
    """The C++ scene constructors (raytracing_amd/host/examples.hpp, through rth_example) and
    raytracing_amd/scenes.py are two statements of the same example files: same surfaces in the
    same order, bit for bit, and the same camera."""
    surfaces, cfg = rt.example(name)
    want, cam = S.EXAMPLES[name]()
    assert len(surfaces) == len(want)
    for field in rt.SURFACE_DTYPE.names:
        a, b = np.ascontiguousarray(surfaces[field]), np.ascontiguousarray(want[field])
        if a.dtype == np.float64:
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), field
        else:
            assert np.array_equal(a, b), field
    wcfg = rt.camera_config(cam)
    assert bytes(cfg) == bytes(wcfg)


def test_cpp_example_errors_and_seed(rt):
    with pytest.raises(rt.RtxError):
        rt.example("no_such_example")
    a, _ = rt.example("cover_spheres", 1)
    b, _ = rt.example("cover_spheres", 2)
    w, _ = S.cover_spheres(seed=2)
    assert not np.array_equal(a["q"][:10], b["q"][:10])
    assert np.array_equal(b["q"], w["q"])


def test_example_program_lists_the_examples():
    import subprocess

    exe = os.path.join(ROOT, "bin", "rtx_example")
    if not os.path.exists(exe):
        pytest.skip("bin/rtx_example is not built (run make)")
    out = subprocess.run([exe, "--list"], capture_output=True, text=True, check=True).stdout.split()
    assert out == list(S.EXAMPLES)
    assert subprocess.run([exe, "nope"], capture_output=True).returncode == 1


def test_rust_bindings_declare_the_whole_abi():
    """bindings/rust/rtx-sys mirrors include/rtx.h: every function, and the struct fields in the header's order
    (the crate cannot be compiled here; this keeps its text in step with the header)."""
    header = open(os.path.join(ROOT, "include", "rtx.h")).read()
    rust = open(os.path.join(ROOT, "bindings", "rust", "rtx-sys", "src", "lib.rs")).read()
    for name in declared_functions("rtx.h"):
        assert re.search(r"pub fn %s\s*\(" % name, rust), f"{name} missing from rtx-sys"
    for struct in ("rtx_scene_desc", "rtx_camera", "rtx_stats", "rtx_build_stats", "rtx_multi_stats"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        c_fields = re.findall(r"(\w+)(?:\[\w+\])?\s*;", body)
        r_body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, rust, re.S).group(1)
        r_fields = re.findall(r"pub (\w+)\s*:", r_body)
        assert c_fields == r_fields, (struct, c_fields, r_fields)
