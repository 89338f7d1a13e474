"""Scene and camera data of the reference's example programs, restated as data.

Each function returns ``(surfaces, camera)``: ``surfaces`` is a numpy array of
``SURFACE_DTYPE`` records (the ``rtx_surface`` struct of include/rtx_host.h, one
record per ``Surface::new(geometry, material)`` call in the example, in the
same order), ``camera`` a dict with the fields of ``struct Camera``
(src/camera.rs:11-26; unspecified fields take the defaults of :45-61).

Citations are into the reference repository seridescent/raytracing.
"""
import math

import numpy as np

SPHERE, QUAD, TRIANGLE = 0, 1, 2
LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT, UV_GRADIENT = 0, 1, 2, 3, 4

SURFACE_DTYPE = np.dtype(
    [
        ("geom_kind", "<u4"),
        ("mat_kind", "<u4"),
        ("q", "<f8", (3,)),
        ("u", "<f8", (3,)),
        ("v", "<f8", (3,)),
        ("radius", "<f8"),
        ("rgb", "<f8", (3,)),
        ("param", "<f8"),
    ],
    align=True,
)
assert SURFACE_DTYPE.itemsize == 120

CAMERA_DEFAULTS = dict(  # src/camera.rs:45-61
    aspect_ratio=1.0,
    image_width=100,
    samples_per_pixel=10,
    max_depth=10,
    v_fov=90.0,
    look_from=(0.0, 0.0, 0.0),
    look_at=(0.0, 0.0, -1.0),
    v_up=(0.0, 1.0, 0.0),
    defocus_angle=0.0,
    focus_dist=10.0,
    background=(0.0, 0.0, 0.0),
)


def camera(**kw):
    c = dict(CAMERA_DEFAULTS)
    c.update(kw)
    return c


def lambertian(albedo):
    return (LAMBERTIAN, tuple(albedo), 0.0)


def metal(albedo, fuzz_radius):
    return (METAL, tuple(albedo), float(fuzz_radius))


def dielectric(refraction_index):
    return (DIELECTRIC, (0.0, 0.0, 0.0), float(refraction_index))


def diffuse_light(emit):
    return (DIFFUSE_LIGHT, tuple(emit), 0.0)


def uv_gradient(intensity):
    return (UV_GRADIENT, (0.0, 0.0, 0.0), float(intensity))


def sphere(center, radius, material):
    s = np.zeros((), SURFACE_DTYPE)
    s["geom_kind"] = SPHERE
    s["q"] = center
    s["radius"] = radius
    s["mat_kind"], s["rgb"], s["param"] = material
    return s


def _planar(kind, q, u, v, material):
    s = np.zeros((), SURFACE_DTYPE)
    s["geom_kind"] = kind
    s["q"], s["u"], s["v"] = q, u, v
    s["mat_kind"], s["rgb"], s["param"] = material
    return s


def quad(q, u, v, material):
    return _planar(QUAD, q, u, v, material)


def triangle(q, u, v, material):
    return _planar(TRIANGLE, q, u, v, material)


def pack(surfaces):
    out = np.zeros(len(surfaces), SURFACE_DTYPE)
    for i, s in enumerate(surfaces):
        out[i] = s
    return out


class SplitMix64:
    """Seeded stand-in for rand's OS-seeded ThreadRng in scene generation."""

    def __init__(self, seed):
        self.state = seed & 0xFFFFFFFFFFFFFFFF

    def next_u64(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def random(self):  # random::<f64>()
        return (self.next_u64() >> 11) * (2.0 ** -53)

    def random_range(self, lo, hi):  # random_range(lo..hi)
        return self.random() * (hi - lo) + lo


def hello_triangle():
    """examples/hello_triangle.rs:13-51"""
    side_length = 2.0
    height = side_length * math.sqrt(3.0) / 2.0
    top = (0.0, height * 0.5, 0.0)
    bottom_left = (-side_length * 0.5, -height * 0.5, 0.0)
    bottom_right = (side_length * 0.5, -height * 0.5, 0.0)
    u = tuple(a - b for a, b in zip(bottom_right, bottom_left))
    v = tuple(a - b for a, b in zip(top, bottom_left))
    surfaces = pack([triangle(bottom_left, u, v, uv_gradient(1.0))])
    cam = camera(
        aspect_ratio=16.0 / 9.0,
        image_width=1920,
        samples_per_pixel=1,
        max_depth=50,
        v_fov=45.0,
        look_from=(0.0, 0.0, 3.0),
        look_at=(0.0, 0.0, 0.0),
        v_up=(0.0, 1.0, 0.0),
    )
    return surfaces, cam


def cover_spheres(seed=0xC0FFEE, image_width=1200, samples_per_pixel=500):
    """examples/cover_spheres.rs:12-133.  The example draws its scene from the
    OS-seeded RNG; here the same procedure, in the same draw order, runs on a
    seeded generator so every run (and the CPU oracle) sees the same scene."""
    rng = SplitMix64(seed)
    small_r, big_r = 0.2, 1.0
    world = [sphere((0.0, -1000.0, 0.0), 1000.0, lambertian((0.5, 0.5, 0.5)))]
    big = [
        sphere((-4.0, 1.0, 0.0), big_r, lambertian((0.4, 0.2, 0.1))),
        sphere((0.0, 1.0, 0.0), big_r, dielectric(1.5)),
        sphere((4.0, 1.0, 0.0), big_r, metal((0.7, 0.6, 0.5), 0.0)),
    ]
    for a in range(-11, 11):
        for b in range(-11, 11):
            cx = float(a) + 0.9 * rng.random()
            cz = float(b) + 0.9 * rng.random()
            center = (cx, small_r, cz)
            too_close = False
            for s in big:
                d = [float(s["q"][i]) - center[i] for i in range(3)]
                if math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < big_r + small_r:
                    too_close = True
                    break
            if too_close:
                continue
            choose = rng.random()
            if choose < 0.8:
                v1 = [rng.random() for _ in range(3)]
                v2 = [rng.random() for _ in range(3)]
                mat = lambertian([x * y for x, y in zip(v1, v2)])
            elif choose < 0.95:
                albedo = [rng.random_range(0.5, 1.0) for _ in range(3)]
                mat = metal(albedo, rng.random_range(0.0, 0.5))
            else:
                mat = dielectric(1.5)
            world.append(sphere(center, small_r, mat))
    world.extend(big)
    cam = camera(
        aspect_ratio=16.0 / 9.0,
        image_width=image_width,
        samples_per_pixel=samples_per_pixel,
        max_depth=50,
        look_from=(13.0, 2.0, 3.0),
        look_at=(0.0, 0.0, 0.0),
        v_fov=20.0,
        defocus_angle=0.6,
        focus_dist=10.0,
        background=(0.7, 0.8, 1.0),
    )
    return pack(world), cam


def _box_geometry(a, b, material, theta):
    """examples/cornell_box.rs:67-130"""
    mn = [min(a[i], b[i]) for i in range(3)]
    mx = [max(a[i], b[i]) for i in range(3)]
    center = [mn[i] + (mx[i] - mn[i]) * 0.5 for i in range(3)]
    cos_t, sin_t = math.cos(theta), math.sin(theta)

    def rotate_y(v):
        rel = [v[i] - center[i] for i in range(3)]
        rot = (cos_t * rel[0] + sin_t * rel[2], rel[1], -sin_t * rel[0] + cos_t * rel[2])
        return tuple(rot[i] + center[i] for i in range(3))

    v000 = rotate_y((mn[0], mn[1], mn[2]))
    v001 = rotate_y((mn[0], mn[1], mx[2]))
    v010 = rotate_y((mn[0], mx[1], mn[2]))
    v011 = rotate_y((mn[0], mx[1], mx[2]))
    v100 = rotate_y((mx[0], mn[1], mn[2]))
    v101 = rotate_y((mx[0], mn[1], mx[2]))
    v110 = rotate_y((mx[0], mx[1], mn[2]))
    v111 = rotate_y((mx[0], mx[1], mx[2]))

    def sub(p, q):
        return tuple(p[i] - q[i] for i in range(3))

    return [
        quad(v001, sub(v101, v001), sub(v011, v001), material),  # front
        quad(v100, sub(v000, v100), sub(v110, v100), material),  # back
        quad(v000, sub(v001, v000), sub(v010, v000), material),  # left
        quad(v101, sub(v100, v101), sub(v111, v101), material),  # right
        quad(v000, sub(v100, v000), sub(v001, v000), material),  # bottom
        quad(v010, sub(v011, v010), sub(v110, v010), material),  # top
    ]


def cornell_box(samples_per_pixel=2000, image_width=600):
    """examples/cornell_box.rs:10-202"""
    white = lambertian((0.73, 0.73, 0.73))
    red = lambertian((0.65, 0.05, 0.05))
    green = lambertian((0.12, 0.45, 0.15))
    light = diffuse_light((50.0, 50.0, 50.0))

    def add(p, q):
        return tuple(p[i] + q[i] for i in range(3))

    surfaces = []
    surfaces += _box_geometry(
        add((0.0, 0.0, 0.0), (265.0, 0.0, 295.0)),
        add((165.0, 330.0, 165.0), (265.0, 0.0, 295.0)),
        metal((0.7, 0.6, 0.5), 0.0),
        18.0 * (math.pi / 180.0),
    )
    surfaces += _box_geometry(
        add((0.0, 0.0, 0.0), (100.0, 0.0, 65.0)),
        add((165.0, 165.0, 165.0), (100.0, 0.0, 65.0)),
        white,
        -18.0 * (math.pi / 180.0),
    )
    surfaces += [
        quad((555.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), red),
        quad((0.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), green),
        quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), light),
        quad((0.0, 0.0, 0.0), (555.0, 0.0, 0.0), (0.0, 0.0, 555.0), white),
        quad((555.0, 555.0, 555.0), (-555.0, 0.0, 0.0), (0.0, 0.0, -555.0), white),
        quad((0.0, 0.0, 555.0), (555.0, 0.0, 0.0), (0.0, 555.0, 0.0), white),
    ]
    cam = camera(
        aspect_ratio=1.0,
        image_width=image_width,
        samples_per_pixel=samples_per_pixel,
        max_depth=50,
        v_fov=40.0,
        look_from=(278.0, 278.0, -800.0),
        look_at=(278.0, 278.0, 0.0),
        v_up=(0.0, 1.0, 0.0),
        background=(0.0, 0.0, 0.0),
    )
    return pack(surfaces), cam


def quads():
    """examples/quads.rs:13-98"""
    surfaces = pack(
        [
            quad((-3.0, -2.0, 5.0), (0.0, 0.0, -4.0), (0.0, 4.0, 0.0), lambertian((1.0, 0.2, 0.2))),
            quad((-2.0, -2.0, 0.0), (4.0, 0.0, 0.0), (0.0, 4.0, 0.0), lambertian((0.2, 1.0, 0.2))),
            quad((3.0, -2.0, 1.0), (0.0, 0.0, 4.0), (0.0, 4.0, 0.0), lambertian((0.2, 0.2, 1.0))),
            quad((-2.0, 3.0, 1.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), lambertian((1.0, 0.5, 0.0))),
            quad((-2.0, -3.0, 5.0), (4.0, 0.0, 0.0), (0.0, 0.0, -4.0), lambertian((0.2, 0.8, 0.8))),
        ]
    )
    cam = camera(
        aspect_ratio=1.0,
        image_width=400,
        samples_per_pixel=100,
        max_depth=50,
        v_fov=80.0,
        look_from=(0.0, 0.0, 9.0),
        look_at=(0.0, 0.0, 0.0),
        v_up=(0.0, 1.0, 0.0),
        background=(0.7, 0.8, 1.0),
    )
    return surfaces, cam


def simple_light():
    """examples/simple_light.rs:13-70"""
    surfaces = pack(
        [
            sphere((0.0, -1000.0, 0.0), 1000.0, lambertian((0.6, 0.5, 0.4))),
            sphere((0.0, 2.0, 0.0), 2.0, lambertian((0.8, 0.4, 0.6))),
            quad((3.0, 1.0, -2.0), (2.0, 0.0, 0.0), (0.0, 2.0, 0.0), diffuse_light((10.0, 10.0, 10.0))),
        ]
    )
    cam = camera(
        aspect_ratio=16.0 / 9.0,
        image_width=400,
        samples_per_pixel=1000,
        max_depth=50,
        v_fov=20.0,
        look_from=(26.0, 3.0, 6.0),
        look_at=(0.0, 2.0, 0.0),
        v_up=(0.0, 1.0, 0.0),
        focus_dist=1.0,
        background=(0.0, 0.0, 0.0),
    )
    return surfaces, cam


def simple_scene():
    """examples/simple_scene.rs:13-70"""
    surfaces = pack(
        [
            sphere((0.0, 0.0, -1.0), 0.5, lambertian((0.7, 0.3, 0.3))),
            sphere((-1.0, 0.0, -1.0), 0.5, lambertian((0.3, 0.3, 0.7))),
            sphere((1.0, 0.0, -1.0), 0.5, metal((0.8, 0.8, 0.9), 0.0)),
            sphere((0.0, -100.5, -1.0), 100.0, lambertian((0.8, 0.8, 0.0))),
        ]
    )
    cam = camera(
        aspect_ratio=16.0 / 9.0,
        image_width=400,
        samples_per_pixel=100,
        max_depth=50,
        v_fov=90.0,
        look_from=(0.0, 0.0, 0.0),
        look_at=(0.0, 0.0, -1.0),
        v_up=(0.0, 1.0, 0.0),
        focus_dist=1.0,
        background=(0.7, 0.8, 1.0),
    )
    return surfaces, cam


def demo_spheres():
    """examples/demo_spheres.rs:10-79"""
    surfaces = pack(
        [
            sphere((0.0, -100.5, -1.0), 100.0, lambertian((0.8, 0.8, 0.0))),
            sphere((0.0, 0.0, -1.2), 0.5, lambertian((0.1, 0.2, 0.5))),
            sphere((-1.0, 0.0, -1.0001), 0.5, dielectric(1.5)),
            sphere((-1.0, 0.0, -1.0), 0.4, dielectric(1.0 / 1.5)),
            sphere((1.0, 0.0, -1.0), 0.5, metal((0.8, 0.6, 0.2), 1.0)),
        ]
    )
    cam = camera(
        aspect_ratio=16.0 / 9.0,
        image_width=400,
        samples_per_pixel=100,
        max_depth=50,
        look_from=(-2.0, 2.0, 1.0),
        look_at=(0.0, 0.0, -1.0),
        v_fov=20.0,
        defocus_angle=10.0,
        focus_dist=3.4,
        background=(0.7, 0.8, 1.0),
    )
    return surfaces, cam


def synthetic_soup(n=1_000_000, kind="triangles", seed=42, image_width=3840, samples_per_pixel=1000):
    """The bandwidth-stress scene (not in the reference): `n` random triangles
    (q ~ U[-50,50)^3, u,v ~ U[-0.5,0.5)^3, Lambertian U[0.1,0.9)^3, 1 % lights)
    or `n` random spheres (centre U[-50,50)^3, r U[0.05,0.3), 80/15/5 %
    Lambertian / Metal / Dielectric as in cover_spheres)."""
    rng = np.random.default_rng(seed)
    out = np.zeros(n, SURFACE_DTYPE)
    if kind == "triangles":
        out["geom_kind"] = TRIANGLE
        out["q"] = rng.uniform(-50.0, 50.0, (n, 3))
        out["u"] = rng.uniform(-0.5, 0.5, (n, 3))
        out["v"] = rng.uniform(-0.5, 0.5, (n, 3))
        out["mat_kind"] = LAMBERTIAN
        out["rgb"] = rng.uniform(0.1, 0.9, (n, 3))
        lights = rng.random(n) < 0.01
        out["mat_kind"][lights] = DIFFUSE_LIGHT
        out["rgb"][lights] = (5.0, 5.0, 5.0)
    else:
        out["geom_kind"] = SPHERE
        out["q"] = rng.uniform(-50.0, 50.0, (n, 3))
        out["radius"] = rng.uniform(0.05, 0.3, n)
        choose = rng.random(n)
        lam = choose < 0.8
        met = (choose >= 0.8) & (choose < 0.95)
        die = choose >= 0.95
        out["mat_kind"][lam] = LAMBERTIAN
        out["rgb"][lam] = rng.random((int(lam.sum()), 3)) * rng.random((int(lam.sum()), 3))
        out["mat_kind"][met] = METAL
        out["rgb"][met] = rng.uniform(0.5, 1.0, (int(met.sum()), 3))
        out["param"][met] = rng.uniform(0.0, 0.5, int(met.sum()))
        out["mat_kind"][die] = DIELECTRIC
        out["param"][die] = 1.5
    cam = camera(
        aspect_ratio=16.0 / 9.0,
        image_width=image_width,
        samples_per_pixel=samples_per_pixel,
        max_depth=50,
        v_fov=40.0,
        look_from=(0.0, 0.0, -160.0),
        look_at=(0.0, 0.0, 0.0),
        background=(0.7, 0.8, 1.0),
    )
    return out, cam


EXAMPLES = {
    "hello_triangle": hello_triangle,
    "cover_spheres": cover_spheres,
    "cornell_box": cornell_box,
    "quads": quads,
    "simple_light": simple_light,
    "simple_scene": simple_scene,
    "demo_spheres": demo_spheres,
}
