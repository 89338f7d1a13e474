"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

NONE = 0xFFFFFFFF
BISECT, MIDPOINT, SAH_EQUAL_SIZE, SAH_PER_SURFACE = 0, 1, 2, 3

SURFACE_DTYPE = np.dtype(
    [("geom_kind", "<u4"), ("mat_kind", "<u4"), ("q", "<f8", (3,)), ("u", "<f8", (3,)), ("v", "<f8", (3,)),
     ("radius", "<f8"), ("rgb", "<f8", (3,)), ("param", "<f8")],
    align=True,
)


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


if not os.path.exists(LIB_PATH):
    build()
_lib = C.CDLL(LIB_PATH)
_p = C.POINTER
_dp, _u32p, _u8p = _p(C.c_double), _p(C.c_uint32), _p(C.c_uint8)


class CameraConfig(C.Structure):  # orc_camera
    _fields_ = [
        ("aspect_ratio", C.c_double), ("image_width", C.c_uint32), ("samples_per_pixel", C.c_uint32),
        ("max_depth", C.c_uint32), ("_pad", C.c_uint32), ("v_fov", C.c_double),
        ("look_from", C.c_double * 3), ("look_at", C.c_double * 3), ("v_up", C.c_double * 3),
        ("defocus_angle", C.c_double), ("focus_dist", C.c_double), ("background", C.c_double * 3),
    ]


class ICamera(C.Structure):  # orc_icamera
    _fields_ = [
        ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
        ("defocus_angle", C.c_double), ("background", C.c_double * 3), ("pixel_samples_scale", C.c_double),
        ("center", C.c_double * 3), ("pixel00_loc", C.c_double * 3), ("pixel_du", C.c_double * 3), ("pixel_dv", C.c_double * 3),
        ("defocus_disk_u", C.c_double * 3), ("defocus_disk_v", C.c_double * 3),
    ]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("node_visits", C.c_uint64), ("sphere_tests", C.c_uint64), ("planar_tests", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib.orc_last_error.restype = C.c_char_p
_lib.orc_scene_create.restype = C.c_void_p
_lib.orc_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
_lib.orc_scene_destroy.argtypes = [C.c_void_p]
_lib.orc_scene_destroy.restype = None
_lib.orc_scene_n_nodes.argtypes = [C.c_void_p]
_lib.orc_scene_n_nodes.restype = C.c_uint32
_lib.orc_scene_nodes.argtypes = [C.c_void_p, _dp, _dp, _u32p, _u32p]
_lib.orc_scene_nodes.restype = None
_lib.orc_surface_bbox.argtypes = [C.c_void_p, _dp]
_lib.orc_sah_cost.argtypes = [_dp, C.c_uint64, _dp, C.c_uint64, _dp]
_lib.orc_sah_cost.restype = C.c_double
_lib.orc_aabb_hit.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double]
_lib.orc_surface_hit.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_double, _dp]
_lib.orc_trace_closest.argtypes = [C.c_void_p, _dp, _dp, C.c_uint64, C.c_double, C.c_double, _u32p, _dp, _p(Counters)]
_lib.orc_trace_closest.restype = None
_lib.orc_trace_bruteforce.argtypes = [C.c_void_p, _dp, _dp, C.c_uint64, C.c_double, C.c_double, _u32p, _dp]
_lib.orc_trace_bruteforce.restype = None
_lib.orc_camera_initialize.argtypes = [_p(CameraConfig), _p(ICamera)]
_lib.orc_camera_initialize.restype = None
_lib.orc_primary_rays.argtypes = [_p(ICamera), C.c_uint64, C.c_uint32, _dp, _dp]
_lib.orc_primary_rays.restype = None
_lib.orc_render.argtypes = [C.c_void_p, _p(ICamera), C.c_uint64, C.c_uint32, C.c_uint32, _dp, C.c_int, C.c_int, _p(Counters)]
_lib.orc_render.restype = None
_lib.orc_ppm_quantize.argtypes = [_dp, C.c_uint64, _u8p]
_lib.orc_ppm_quantize.restype = None
_lib.orc_philox4x32.argtypes = [_u32p, _u32p, C.c_int, _u32p]
_lib.orc_philox4x32.restype = None
_lib.orc_philox_rounds.argtypes = []
_lib.orc_philox_rounds.restype = C.c_int
PHILOX_ROUNDS = _lib.orc_philox_rounds()
_lib.orc_rng_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _dp]
_lib.orc_rng_stream.restype = None
_lib.orc_scatter.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _dp]
_lib.orc_scatter.restype = None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class OracleError(RuntimeError):
    pass


def camera_config(cam):
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
    cfg = cam if isinstance(cam, CameraConfig) else camera_config(cam)
    out = ICamera()
    _lib.orc_camera_initialize(C.byref(cfg), C.byref(out))
    return out


class Scene:
    def __init__(self, surfaces, strategy=SAH_PER_SURFACE, buckets=0):
        self.surfaces = np.ascontiguousarray(surfaces, dtype=SURFACE_DTYPE)
        self._h = _lib.orc_scene_create(self.surfaces.ctypes.data_as(C.c_void_p), len(self.surfaces), strategy, buckets)
        if not self._h:
            raise OracleError(_lib.orc_last_error().decode())

    @property
    def n_nodes(self):
        return int(_lib.orc_scene_n_nodes(self._h))

    def node_tables(self):
        n = self.n_nodes
        bmin = np.zeros((3, n)); bmax = np.zeros((3, n))
        right = np.zeros(n, np.uint32); leaf = np.zeros(n, np.uint32)
        if n:
            _lib.orc_scene_nodes(self._h, _ptr(bmin, _dp), _ptr(bmax, _dp), _ptr(right, _u32p), _ptr(leaf, _u32p))
        return bmin, bmax, right, leaf

    def trace_closest(self, org, direction, tmin=0.001, tmax=float("inf")):
        org, direction = _f64(org), _f64(direction)
        n = org.shape[0]
        ids = np.empty(n, np.uint32); t = np.empty(n, np.float64)
        ctr = Counters()
        _lib.orc_trace_closest(self._h, _ptr(org, _dp), _ptr(direction, _dp), n, tmin, tmax, _ptr(ids, _u32p), _ptr(t, _dp), C.byref(ctr))
        return ids, t, ctr

    def trace_bruteforce(self, org, direction, tmin=0.001, tmax=float("inf")):
        org, direction = _f64(org), _f64(direction)
        n = org.shape[0]
        ids = np.empty(n, np.uint32); t = np.empty(n, np.float64)
        _lib.orc_trace_bruteforce(self._h, _ptr(org, _dp), _ptr(direction, _dp), n, tmin, tmax, _ptr(ids, _u32p), _ptr(t, _dp))
        return ids, t

    def render(self, icam, seed, spp_begin, spp_count, n_threads=0, iterative=False):
        """Unscaled radiance sums [H,W,3] over the sample range, and counters."""
        out = np.zeros((icam.image_height, icam.image_width, 3))
        ctr = Counters()
        _lib.orc_render(self._h, C.byref(icam), seed, spp_begin, spp_count, _ptr(out, _dp), n_threads, 1 if iterative else 0, C.byref(ctr))
        return out, ctr

    def close(self):
        if self._h:
            _lib.orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def surface_bbox(surface):
    s = np.ascontiguousarray(surface, dtype=SURFACE_DTYPE).reshape(1)
    out = np.zeros(6)
    if _lib.orc_surface_bbox(s.ctypes.data_as(C.c_void_p), _ptr(out, _dp)) != 0:
        raise OracleError(_lib.orc_last_error().decode())
    return out


def sah_cost(left, n_left, right, n_right, parent):
    l, r, p = _f64(left), _f64(right), _f64(parent)
    return float(_lib.orc_sah_cost(_ptr(l, _dp), n_left, _ptr(r, _dp), n_right, _ptr(p, _dp)))


def aabb_hit(box, org, direction, tmin, tmax):
    b, o, d = _f64(box), _f64(org), _f64(direction)
    return bool(_lib.orc_aabb_hit(_ptr(b, _dp), _ptr(o, _dp), _ptr(d, _dp), tmin, tmax))


def surface_hit(surface, org, direction, tmin=0.001, tmax=float("inf")):
    """None on a miss, else dict(t, p, alpha, beta, front_face, normal)."""
    s = np.ascontiguousarray(surface, dtype=SURFACE_DTYPE).reshape(1)
    o, d = _f64(org), _f64(direction)
    out = np.zeros(10)
    rc = _lib.orc_surface_hit(s.ctypes.data_as(C.c_void_p), _ptr(o, _dp), _ptr(d, _dp), tmin, tmax, _ptr(out, _dp))
    if rc < 0:
        raise OracleError(_lib.orc_last_error().decode())
    if rc == 0:
        return None
    return dict(t=out[0], p=out[1:4].copy(), alpha=out[4], beta=out[5], front_face=bool(out[6]), normal=out[7:10].copy(), raw=out)


def primary_rays(icam, seed, sample):
    n = icam.image_width * icam.image_height
    org = np.empty((n, 3)); direction = np.empty((n, 3))
    _lib.orc_primary_rays(C.byref(icam), seed, sample, _ptr(org, _dp), _ptr(direction, _dp))
    return org, direction


def ppm_quantize(rgb):
    rgb = _f64(rgb)
    out = np.empty(rgb.shape, np.uint8)
    _lib.orc_ppm_quantize(_ptr(rgb, _dp), rgb.size // 3, _ptr(out, _u8p))
    return out


def philox4x32(ctr, key, rounds=PHILOX_ROUNDS):
    c = np.asarray(ctr, np.uint32); k = np.asarray(key, np.uint32); o = np.zeros(4, np.uint32)
    _lib.orc_philox4x32(_ptr(c, _u32p), _ptr(k, _u32p), int(rounds), _ptr(o, _u32p))
    return o


def rng_stream(seed, pixel, sample, n):
    out = np.empty(n)
    _lib.orc_rng_stream(seed, pixel, sample, n, _ptr(out, _dp))
    return out


def scatter(surface, org, direction, hit_raw, seed, pixel, sample, draw):
    s = np.ascontiguousarray(surface, dtype=SURFACE_DTYPE).reshape(1)
    o, d, h = _f64(org), _f64(direction), _f64(hit_raw)
    out = np.zeros(14)
    _lib.orc_scatter(s.ctypes.data_as(C.c_void_p), _ptr(o, _dp), _ptr(d, _dp), _ptr(h, _dp), seed, pixel, sample, draw, _ptr(out, _dp))
    return dict(scattered=bool(out[0]), origin=out[1:4].copy(), direction=out[4:7].copy(), attenuation=out[7:10].copy(),
                emitted=out[10:13].copy(), draw=int(out[13]))
