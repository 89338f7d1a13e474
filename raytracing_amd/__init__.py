"""raytracing_amd -- MI355X render backend for seridescent/raytracing's
sample/bounce loop, plus the host-side scene / camera / BVH-builder API.

This module is a thin ctypes binding of ``librtx.so`` (the C ABI declared in
``include/rtx.h`` and ``include/rtx_host.h``).  All compute runs in the HIP
kernels behind that ABI; there is no CPU fallback: if the library is missing
the import fails.
"""
import ctypes as C
import os

import numpy as np

try:
    # PyTorch ships its own copy of the HIP runtime under the same soname as
    # the system one.  Loading it first makes librtx.so bind to that same copy,
    # so device pointers and streams can be passed between torch and this
    # library.  torch is only plumbing here; without it the system runtime is used.
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

from . import scenes
from .scenes import SURFACE_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtx.so")

NONE = 0xFFFFFFFF
FLAG_COUNT_WORK = 1
FLAG_TIME_KERNELS = 2
FLAG_EXACT_DIVIDE = 4
FLAG_F64_FILTER = 8
FLAG_NO_LDS = 16
FLAG_NO_WAIT = 32
FLAG_COUNT_ACTUAL = 64
FORM_NONE, FORM_ROUND_LOOP, FORM_POOLED, FORM_RESIDENT = 0, 1, 2, 3
FORM_KERNEL = {FORM_ROUND_LOOP: "k_generate + k_traverse_f32 + k_shade", FORM_POOLED: "k_render_persistent", FORM_RESIDENT: "k_render_resident"}
TRACE_DEVICE_MAX_RAYS = 1 << 18

BISECT, MIDPOINT, SAH_EQUAL_SIZE, SAH_PER_SURFACE = 0, 1, 2, 3


class RtxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtx error {code}: {message}")
        self.code = code


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "raytracing_amd has no CPU fallback."
    )
_lib = C.CDLL(LIB_PATH)

_p = C.POINTER
_dp, _u32p, _u8p = _p(C.c_double), _p(C.c_uint32), _p(C.c_uint8)


class SceneDesc(C.Structure):  # rtx_scene_desc
    _fields_ = [
        ("n_nodes", C.c_uint32), ("n_prims", C.c_uint32), ("n_materials", C.c_uint32), ("_reserved", C.c_uint32),
        ("node_bmin", _dp), ("node_bmax", _dp), ("node_right", _u32p), ("node_prim", _u32p),
        ("prim_kind", _u8p), ("prim_q", _dp), ("prim_u", _dp), ("prim_v", _dp), ("prim_norm", _dp), ("prim_w", _dp),
        ("prim_d", _dp), ("prim_radius", _dp), ("prim_material", _u32p), ("prim_user_id", _u32p),
        ("mat_kind", _u32p), ("mat_rgb", _dp), ("mat_param", _dp),
    ]


class Camera(C.Structure):  # rtx_camera == InitializedCamera
    _fields_ = [
        ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
        ("defocus_angle", C.c_double), ("background", C.c_double * 3), ("pixel_samples_scale", C.c_double),
        ("center", C.c_double * 3), ("pixel00_loc", C.c_double * 3), ("pixel_du", C.c_double * 3), ("pixel_dv", C.c_double * 3),
        ("defocus_disk_u", C.c_double * 3), ("defocus_disk_v", C.c_double * 3),
    ]


class CameraConfig(C.Structure):  # rtx_camera_config == struct Camera
    _fields_ = [
        ("aspect_ratio", C.c_double), ("image_width", C.c_uint32), ("samples_per_pixel", C.c_uint32),
        ("max_depth", C.c_uint32), ("_reserved", C.c_uint32), ("v_fov", C.c_double),
        ("look_from", C.c_double * 3), ("look_at", C.c_double * 3), ("v_up", C.c_double * 3),
        ("defocus_angle", C.c_double), ("focus_dist", C.c_double), ("background", C.c_double * 3),
    ]


class Stats(C.Structure):  # rtx_stats
    _fields_ = [
        ("rays", C.c_uint64), ("paths", C.c_uint64), ("node_visits", C.c_uint64), ("sphere_tests", C.c_uint64),
        ("planar_tests", C.c_uint64), ("box_fallbacks", C.c_uint64), ("iterations", C.c_uint64), ("ms_total", C.c_double), ("ms_traverse", C.c_double),
        ("ms_shade", C.c_double), ("ms_generate", C.c_double), ("kernel_form", C.c_uint32), ("_reserved", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


MULTI_MAX_GPUS = 16


class MultiStats(C.Structure):  # rtx_multi_stats
    _fields_ = [
        ("rays", C.c_uint64), ("paths", C.c_uint64), ("n_gpus", C.c_uint32), ("collective_ranks", C.c_uint32),
        ("ms_frame", C.c_double), ("ms_reduce", C.c_double),
        ("spp_begin", C.c_uint32 * MULTI_MAX_GPUS), ("spp_count", C.c_uint32 * MULTI_MAX_GPUS),
        ("ms_render", C.c_double * MULTI_MAX_GPUS), ("ms_device", C.c_double * MULTI_MAX_GPUS),
    ]


class BuildStats(C.Structure):  # rtx_build_stats
    _fields_ = [
        ("levels", C.c_uint32), ("n_nodes", C.c_uint32), ("ms_sort", C.c_double), ("ms_levels", C.c_double),
        ("device_bytes", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/rtx.h and include/rtx_host.h declare
ABI_SYMBOLS = [
    "rtx_last_error", "rtx_device_count", "rtx_scene_create", "rtx_scene_destroy", "rtx_trace_closest", "rtx_trace_closest_device",
    "rtx_primary_rays", "rtx_render_accumulate", "rtx_scale", "rtx_quantize", "rtx_render", "rtx_selftest_math",
    "rtx_selftest_rng", "rtx_selftest_cull_boxes", "rtx_bvh_build", "rtx_multi_create", "rtx_multi_destroy", "rtx_multi_render", "rtx_multi_image", "rtx_render_multi", "rtx_ppm_text_bound", "rtx_ppm_text", "rtx_render_ppm",
    "rth_last_error", "rth_bvh_build", "rth_bvh_build_device", "rth_run_device_build", "rth_example", "rth_bvh_desc", "rth_bvh_destroy", "rth_camera_initialize", "rth_run", "rth_run_multi",
]

_lib.rtx_last_error.restype = C.c_char_p
_lib.rth_last_error.restype = C.c_char_p
_lib.rtx_device_count.restype = C.c_int
_lib.rtx_scene_create.argtypes = [_p(SceneDesc), _p(C.c_void_p)]
_lib.rtx_scene_destroy.argtypes = [C.c_void_p]
_lib.rtx_scene_destroy.restype = None
_lib.rtx_trace_closest.argtypes = [C.c_void_p, _dp, _dp, C.c_uint64, C.c_double, C.c_double, C.c_uint32, _u32p, _dp, _p(Stats)]
_lib.rtx_trace_closest_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.rtx_primary_rays.argtypes = [_p(Camera), C.c_uint64, C.c_uint32, _dp, _dp]
_lib.rtx_render_accumulate.argtypes = [C.c_void_p, _p(Camera), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, _p(Stats)]
_lib.rtx_scale.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p]
_lib.rtx_quantize.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
_lib.rtx_render.argtypes = [C.c_void_p, _p(Camera), C.c_uint64, C.c_uint32, _dp, _p(Stats)]
_lib.rtx_multi_create.argtypes = [_p(SceneDesc), C.c_int, _p(C.c_void_p)]
_lib.rtx_multi_destroy.argtypes = [C.c_void_p]
_lib.rtx_multi_destroy.restype = None
_lib.rtx_multi_render.argtypes = [C.c_void_p, _p(Camera), C.c_uint64, C.c_uint32, _dp, _p(MultiStats)]
_lib.rtx_multi_image.argtypes = [C.c_void_p]
_lib.rtx_multi_image.restype = C.c_void_p
_lib.rtx_render_multi.argtypes = [_p(SceneDesc), _p(Camera), C.c_uint64, C.c_int, C.c_uint32, _dp, _p(MultiStats)]
_lib.rtx_selftest_math.argtypes = [C.c_int, _dp, _dp, C.c_uint64, _dp]
_lib.rtx_selftest_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _dp]
_lib.rtx_selftest_cull_boxes.argtypes = [_p(SceneDesc), C.c_int, _dp, _dp, _u32p]
_lib.rth_bvh_build.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, _p(C.c_void_p)]
_lib.rth_bvh_build_device.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, _p(C.c_void_p), _p(BuildStats)]
_lib.rtx_bvh_build.argtypes = [_dp, C.c_uint32, C.c_int, C.c_uint32, _u32p, _dp, _dp, _u32p, _u32p, _p(BuildStats)]
_lib.rth_run_device_build.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, _p(CameraConfig), C.c_uint64, C.c_char_p, _dp]
_lib.rtx_ppm_text_bound.argtypes = [C.c_uint32, C.c_uint32]
_lib.rtx_ppm_text_bound.restype = C.c_uint64
_lib.rtx_ppm_text.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64, _p(C.c_uint64), C.c_void_p]
_lib.rtx_render_ppm.argtypes = [C.c_void_p, _p(Camera), C.c_uint64, C.c_uint32, _dp, C.c_char_p, C.c_uint64, _p(C.c_uint64), _p(Stats)]
_lib.rth_example.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, _p(C.c_uint32), _p(CameraConfig)]
_lib.rth_bvh_desc.argtypes = [C.c_void_p, _p(SceneDesc)]
_lib.rth_bvh_desc.restype = None
_lib.rth_bvh_destroy.argtypes = [C.c_void_p]
_lib.rth_bvh_destroy.restype = None
_lib.rth_camera_initialize.argtypes = [_p(CameraConfig), _p(Camera)]
_lib.rth_run_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, _p(CameraConfig), C.c_uint64, C.c_int, C.c_char_p, _dp]
_lib.rth_run.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, _p(CameraConfig), C.c_uint64, C.c_char_p, _dp]


def _check(rc, host=False):
    if rc != 0:
        msg = (_lib.rth_last_error() if host else _lib.rtx_last_error()) or b""
        raise RtxError(rc, msg.decode("utf-8", "replace"))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def device_count():
    return int(_lib.rtx_device_count())


def camera_config(cam):
    """dict (fields of ``struct Camera``) -> CameraConfig"""
    c = CameraConfig()
    c.aspect_ratio = cam["aspect_ratio"]
    c.image_width = cam["image_width"]
    c.samples_per_pixel = cam["samples_per_pixel"]
    c.max_depth = cam["max_depth"]
    c.v_fov = cam["v_fov"]
    c.look_from[:] = cam["look_from"]
    c.look_at[:] = cam["look_at"]
    c.v_up[:] = cam["v_up"]
    c.defocus_angle = cam["defocus_angle"]
    c.focus_dist = cam["focus_dist"]
    c.background[:] = cam["background"]
    return c


def camera_initialize(cam):
    """``Camera::initialize`` (src/camera.rs:64-113 of the reference)."""
    cfg = cam if isinstance(cam, CameraConfig) else camera_config(cam)
    out = Camera()
    _check(_lib.rth_camera_initialize(C.byref(cfg), C.byref(out)), host=True)
    return out


def ppm_text(d_rgb_ptr, width, height, stream=None):
    """``rtx_ppm_text``: the P3 text (bytes) of a device-resident [H, W, 3] f64 linear image."""
    cap = int(_lib.rtx_ppm_text_bound(width, height))
    buf = C.create_string_buffer(cap)
    n = C.c_uint64(0)
    _check(_lib.rtx_ppm_text(d_rgb_ptr, width, height, buf, cap, C.byref(n), stream))
    return buf.raw[: n.value]


def bvh_build_device(leaf_boxes, strategy=SAH_PER_SURFACE, buckets=0):
    """``rtx_bvh_build``: leaf boxes [n, 6] -> (leaf_order[n], bmin[3, 2n-1], bmax[3, 2n-1], right, prim, BuildStats)."""
    boxes = _f64(leaf_boxes).reshape(-1, 6)
    n = len(boxes)
    nn = max(2 * n - 1, 0)
    order = np.zeros(n, np.uint32)
    bmin, bmax = np.zeros((3, nn)), np.zeros((3, nn))
    right, prim = np.zeros(nn, np.uint32), np.zeros(nn, np.uint32)
    st = BuildStats()
    _check(_lib.rtx_bvh_build(_ptr(boxes, _dp), n, strategy, buckets, _ptr(order, _u32p), _ptr(bmin, _dp), _ptr(bmax, _dp),
                              _ptr(right, _u32p), _ptr(prim, _u32p), C.byref(st)))
    return order, bmin, bmax, right, prim, st


def example(name, scene_seed=0xC0FFEE):
    """``rth_example``: (surfaces, CameraConfig) of one of the reference's example programs, built by the C++ host API."""
    n = C.c_uint32(0)
    cfg = CameraConfig()
    _check(_lib.rth_example(name.encode(), scene_seed, None, 0, C.byref(n), C.byref(cfg)), host=True)
    surfaces = np.zeros(n.value, SURFACE_DTYPE)
    _check(_lib.rth_example(name.encode(), scene_seed, surfaces.ctypes.data_as(C.c_void_p), n.value, C.byref(n), C.byref(cfg)), host=True)
    return surfaces, cfg


class HostBVH:
    """``BVH::from_slice``: on the CPU (the product's host builder), or with ``device=True`` partitioned by the
    device builder (``rtx_bvh_build``); both give the reference's tree."""

    def __init__(self, surfaces, strategy=SAH_PER_SURFACE, buckets=0, device=False):
        surfaces = np.ascontiguousarray(surfaces, dtype=SURFACE_DTYPE)
        self._h = C.c_void_p()
        self.build_stats = None
        if device:
            self.build_stats = BuildStats()
            _check(_lib.rth_bvh_build_device(surfaces.ctypes.data_as(C.c_void_p), len(surfaces), strategy, buckets, C.byref(self._h),
                                             C.byref(self.build_stats)), host=True)
        else:
            _check(_lib.rth_bvh_build(surfaces.ctypes.data_as(C.c_void_p), len(surfaces), strategy, buckets, C.byref(self._h)), host=True)
        self.desc = SceneDesc()
        _lib.rth_bvh_desc(self._h, C.byref(self.desc))

    def _arr(self, ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    @property
    def n_nodes(self):
        return int(self.desc.n_nodes)

    def node_tables(self):
        """(bmin[3,n], bmax[3,n], right[n], leaf_user_id[n]) of the pre-order node array."""
        n, m = self.desc.n_nodes, self.desc.n_prims
        bmin = self._arr(self.desc.node_bmin, 3 * n, np.float64).reshape(3, n)
        bmax = self._arr(self.desc.node_bmax, 3 * n, np.float64).reshape(3, n)
        right = self._arr(self.desc.node_right, n, np.uint32)
        prim = self._arr(self.desc.node_prim, n, np.uint32)
        user = self._arr(self.desc.prim_user_id, m, np.uint32)
        leaf_user = np.full(n, NONE, np.uint32)
        is_leaf = prim != NONE
        leaf_user[is_leaf] = user[prim[is_leaf]]
        return bmin, bmax, right, leaf_user

    def close(self):
        if self._h:
            _lib.rth_bvh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """A flattened BVH resident in HBM of the current HIP device."""

    def __init__(self, bvh):
        desc = bvh.desc if isinstance(bvh, HostBVH) else bvh
        self._keep = bvh
        self._h = C.c_void_p()
        _check(_lib.rtx_scene_create(C.byref(desc), C.byref(self._h)))

    def trace_closest(self, org, direction, tmin=0.001, tmax=float("inf"), flags=0):
        """``BVH::hit`` for a batch; returns (surface ids, t, stats)."""
        org, direction = _f64(org), _f64(direction)
        n = org.shape[0]
        assert org.shape == (n, 3) and direction.shape == (n, 3)
        ids = np.empty(n, np.uint32)
        t = np.empty(n, np.float64)
        st = Stats()
        _check(_lib.rtx_trace_closest(self._h, _ptr(org, _dp), _ptr(direction, _dp), n, tmin, tmax, flags, _ptr(ids, _u32p), _ptr(t, _dp), C.byref(st)))
        return ids, t, st

    def trace_closest_device(self, d_org_ptr, d_dir_ptr, n, d_out_id_ptr, d_out_t_ptr, tmin=0.001, tmax=float("inf"), flags=0, stream=None):
        """``rtx_trace_closest_device``: the same batch on device buffers, enqueued on `stream`, without any allocation."""
        _check(_lib.rtx_trace_closest_device(self._h, C.c_void_p(d_org_ptr), C.c_void_p(d_dir_ptr), n, tmin, tmax, flags,
                                             C.c_void_p(d_out_id_ptr), C.c_void_p(d_out_t_ptr), C.c_void_p(stream or 0)))

    def render(self, cam, seed=0, flags=0):
        """Full render; returns (image[H,W,3] f64 linear, already x 1/spp, stats)."""
        img = np.empty((cam.image_height, cam.image_width, 3), np.float64)
        st = Stats()
        _check(_lib.rtx_render(self._h, C.byref(cam), seed, flags, _ptr(img, _dp), C.byref(st)))
        return img, st

    def render_accumulate(self, cam, seed, spp_begin, spp_count, d_accum_ptr, stream=None, flags=0):
        """Adds the radiance sums of samples [spp_begin, spp_begin+spp_count) into device memory."""
        st = Stats()
        _check(_lib.rtx_render_accumulate(self._h, C.byref(cam), seed, spp_begin, spp_count, flags, C.c_void_p(d_accum_ptr), C.c_void_p(stream or 0), C.byref(st)))
        return st

    def close(self):
        if self._h:
            _lib.rtx_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScene:
    """``rtx_multi_*``: the scene replicated on the first `n_gpus` devices of this process and their RCCL communicator."""

    def __init__(self, bvh, n_gpus):
        desc = bvh.desc if isinstance(bvh, HostBVH) else bvh
        self._keep = bvh
        self._h = C.c_void_p()
        _check(_lib.rtx_multi_create(C.byref(desc), n_gpus, C.byref(self._h)))

    def render(self, cam, seed=0, flags=0, to_host=True):
        """One frame; returns (image[H,W,3] already x 1/spp or None, MultiStats)."""
        img = np.empty((cam.image_height, cam.image_width, 3), np.float64) if to_host else None
        st = MultiStats()
        _check(_lib.rtx_multi_render(self._h, C.byref(cam), seed, flags, _ptr(img, _dp) if to_host else None, C.byref(st)))
        return img, st

    def image_ptr(self):
        return _lib.rtx_multi_image(self._h)

    def close(self):
        if self._h:
            _lib.rtx_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def primary_rays(cam, seed, sample):
    n = cam.image_width * cam.image_height
    org = np.empty((n, 3), np.float64)
    direction = np.empty((n, 3), np.float64)
    _check(_lib.rtx_primary_rays(C.byref(cam), seed, sample, _ptr(org, _dp), _ptr(direction, _dp)))
    return org, direction


def scale(d_in_ptr, n_values, factor, d_out_ptr, stream=None):
    _check(_lib.rtx_scale(C.c_void_p(d_in_ptr), n_values, factor, C.c_void_p(d_out_ptr), C.c_void_p(stream or 0)))


def quantize(d_in_ptr, n_values, d_out_ptr, stream=None):
    _check(_lib.rtx_quantize(C.c_void_p(d_in_ptr), n_values, C.c_void_p(d_out_ptr), C.c_void_p(stream or 0)))


def selftest_math(op, a, b):
    a, b = _f64(a), _f64(b)
    out = np.empty_like(a)
    _check(_lib.rtx_selftest_math(op, _ptr(a, _dp), _ptr(b, _dp), a.size, _ptr(out, _dp)))
    return out


def cull_boxes(bvh, shrink=True):
    """``rtx_selftest_cull_boxes``: (bmin[3, n], bmax[3, n], nodes whose box is smaller than the reference's) of a HostBVH."""
    desc = bvh.desc if isinstance(bvh, HostBVH) else bvh
    n = int(desc.n_nodes)
    bmin, bmax = np.zeros((3, n)), np.zeros((3, n))
    differ = C.c_uint32(0)
    _check(_lib.rtx_selftest_cull_boxes(C.byref(desc), 1 if shrink else 0, _ptr(bmin, _dp), _ptr(bmax, _dp), C.byref(differ)))
    return bmin, bmax, int(differ.value)


def selftest_rng(seed, pixel, sample, n):
    out = np.empty(n, np.float64)
    _check(_lib.rtx_selftest_rng(seed, pixel, sample, n, _ptr(out, _dp)))
    return out


def run(surfaces, cam, seed=0, strategy=SAH_PER_SURFACE, buckets=0, ppm_path=None):
    """``RenderRunner::run``: build, render, optionally write the P3 image."""
    surfaces = np.ascontiguousarray(surfaces, dtype=SURFACE_DTYPE)
    cfg = cam if isinstance(cam, CameraConfig) else camera_config(cam)
    ic = camera_initialize(cfg)
    img = np.empty((ic.image_height, ic.image_width, 3), np.float64)
    _check(
        _lib.rth_run(surfaces.ctypes.data_as(C.c_void_p), len(surfaces), strategy, buckets, C.byref(cfg), seed,
                     ppm_path.encode() if ppm_path else None, _ptr(img, _dp)),
        host=True,
    )
    return img
