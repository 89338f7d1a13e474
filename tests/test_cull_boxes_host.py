"""The boxes the closest-hit walk tests first (rtx_selftest_cull_boxes: host arithmetic, no GPU) against the two properties
the walk's correctness rests on, checked with the oracle's own box and surface tests:
  (a) every cull box lies inside the reference's box of its node;
  (b) a ray that hits a surface inside [tmin, tmax] passes the reference's box predicate on the cull box of every node
      above that surface -- so a ray that FAILS the predicate on a cull box can have no hit below it.
Plus the structure: spheres keep the reference's box, quads whose reference box does not contain them keep it too, an
internal node has the union of its children's boxes (or the reference's), and planar scenes do get smaller boxes."""
import numpy as np
import pytest

from raytracing_amd import scenes as S

NONE = 0xFFFFFFFF


def mixed_soup(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        q = rng.uniform(-20.0, 20.0, 3)
        mat = S.lambertian((0.5, 0.5, 0.5))
        k = i % 5
        if k == 0:
            out.append(S.sphere(q, rng.uniform(0.2, 1.5), mat))
        elif k == 1:
            out.append(S.triangle(q, rng.uniform(-2, 2, 3), rng.uniform(-2, 2, 3), mat))
        elif k == 2:
            a = rng.integers(0, 3)
            u, v = np.zeros(3), np.zeros(3)
            u[(a + 1) % 3] = rng.uniform(0.5, 3.0)
            v[(a + 2) % 3] = rng.uniform(0.5, 3.0)
            out.append(S.quad(q, u, v, mat))
        elif k == 3:
            s = rng.uniform(0.5, 3.0)
            out.append(S.quad(q, (s, 0.0, s), (-s, 0.0, s), mat))       # sticks out of the reference's two-corner box
        else:
            out.append(S.quad(q, rng.uniform(-2, 2, 3), rng.uniform(-2, 2, 3), mat))
    return S.pack(out)


SCENES = {
    "cornell_box": lambda: S.cornell_box()[0],
    "quads": lambda: S.quads()[0],
    "simple_light": lambda: S.simple_light()[0],
    "hello_triangle": lambda: S.hello_triangle()[0],
    "mixed_soup": lambda: mixed_soup(400, 3),
    "soup_triangles": lambda: S.synthetic_soup(n=2000, kind="triangles", seed=5)[0],
    "cover_spheres": lambda: S.cover_spheres()[0],
}


def tree_of(rt, surfaces):
    bvh = rt.HostBVH(surfaces)
    rmin, rmax, right, leaf_user = bvh.node_tables()
    cmin, cmax, differ = rt.cull_boxes(bvh)
    n = bvh.n_nodes
    parent = np.full(n, NONE, np.uint32)
    for i in range(n):
        if leaf_user[i] == NONE:
            parent[i + 1] = i
            parent[right[i]] = i
    return bvh, rmin, rmax, right, leaf_user, cmin, cmax, differ, parent


@pytest.mark.parametrize("name", list(SCENES))
def test_cull_boxes_structure(rt, name):
    surfaces = SCENES[name]()
    bvh, rmin, rmax, right, leaf_user, cmin, cmax, differ, parent = tree_of(rt, surfaces)
    n = bvh.n_nodes
    assert np.all(cmin >= rmin) and np.all(cmax <= rmax)                              # (a)
    assert differ == int(np.count_nonzero(np.any((cmin != rmin) | (cmax != rmax), axis=0)))
    off_min, off_max, off_differ = rt.cull_boxes(bvh, shrink=False)
    assert off_differ == 0 and np.array_equal(off_min, rmin) and np.array_equal(off_max, rmax)
    planar = surfaces["geom_kind"] != S.SPHERE
    assert (differ > 0) == bool(planar.any())
    for i in range(n):
        if leaf_user[i] != NONE:
            s = surfaces[leaf_user[i]]
            if s["geom_kind"] == S.SPHERE:
                assert np.array_equal(cmin[:, i], rmin[:, i]) and np.array_equal(cmax[:, i], rmax[:, i])
                continue
            corners = [s["q"], s["q"] + s["u"], s["q"] + s["v"]] + ([s["q"] + s["u"] + s["v"]] if s["geom_kind"] == S.QUAD else [])
            lo, hi = np.min(corners, axis=0), np.max(corners, axis=0)
            inside_ref = np.all(lo - rmin[:, i] >= 1e-9) and np.all(rmax[:, i] - hi >= 1e-9)
            if np.array_equal(cmin[:, i], rmin[:, i]) and np.array_equal(cmax[:, i], rmax[:, i]):
                continue                                                              # a reference leaf
            assert inside_ref                                                         # a margin leaf: corners inside with room
            big = np.maximum(np.abs(rmin[:, i]), np.abs(rmax[:, i]))
            need = np.maximum(1e-9 * big, 1e-6 * (np.abs(s["u"]) + np.abs(s["v"])))
            assert np.all(lo - cmin[:, i] >= need) and np.all(cmax[:, i] - hi >= need)
        else:
            umin = np.minimum(cmin[:, i + 1], cmin[:, right[i]])
            umax = np.maximum(cmax[:, i + 1], cmax[:, right[i]])
            is_union = np.array_equal(cmin[:, i], umin) and np.array_equal(cmax[:, i], umax)
            is_ref = np.array_equal(cmin[:, i], rmin[:, i]) and np.array_equal(cmax[:, i], rmax[:, i])
            assert is_union or is_ref
    if name == "mixed_soup":   # the rotated quads stick out of their reference box: they must have kept it
        stuck = 0
        for i in range(n):
            if leaf_user[i] != NONE and leaf_user[i] % 5 == 3:
                assert np.array_equal(cmin[:, i], rmin[:, i]) and np.array_equal(cmax[:, i], rmax[:, i])
                stuck += 1
        assert stuck == 80


@pytest.mark.parametrize("name", ["cornell_box", "quads", "mixed_soup", "soup_triangles"])
def test_a_hit_below_a_node_passes_the_box_predicate_on_its_cull_box(rt, oracle, name):
    """(b), with the oracle's restatement of AABB::hit (src/aabb.rs:55-76) and of the surface tests: rays aimed at the
    surfaces; whenever the oracle's closest hit is surface s at distance t, every node from s's leaf up to the root must pass
    the predicate on its cull box at the upper ends t (the tightest interval the walk can have when it reaches that node with
    s still to be found), 1.5 t and infinity -- outright when the leaf's cull box is built around the surface, and whenever the
    reference's own test of the leaf's box passes when the leaf kept that box."""
    surfaces = SCENES[name]()
    bvh, rmin, rmax, right, leaf_user, cmin, cmax, differ, parent = tree_of(rt, surfaces)
    orc = oracle.Scene(surfaces)
    leaf_of = {int(u): i for i, u in enumerate(leaf_user) if u != NONE}
    rng = np.random.default_rng(9)
    n_rays = 1500
    pick = rng.integers(0, len(surfaces), n_rays)
    s = surfaces[pick]
    a, b = rng.random(n_rays), rng.random(n_rays)
    flip = (s["geom_kind"] == S.TRIANGLE) & (a + b > 1)
    a, b = np.where(flip, 1 - a, a), np.where(flip, 1 - b, b)
    edge = rng.random(n_rays) < 0.3                                   # a share exactly on edges and corners
    a, b = np.where(edge, np.round(a), a), np.where(edge & (rng.random(n_rays) < 0.5), np.round(b), b)
    target = s["q"] + a[:, None] * s["u"] + b[:, None] * s["v"]
    org = target + rng.normal(size=(n_rays, 3)) * rng.choice([0.01, 1.0, 30.0], size=(n_rays, 1))
    d = (target - org) * rng.choice([1e-3, 1.0, 40.0], size=(n_rays, 1))
    wid, wt, _ = orc.trace_closest(org, d)
    checked = implied = 0
    for k in np.flatnonzero(wid != NONE):
        leaf = leaf_of[int(wid[k])]
        margin_leaf = bool(np.any(cmin[:, leaf] != rmin[:, leaf]) or np.any(cmax[:, leaf] != rmax[:, leaf]))
        leaf_ref = np.concatenate([rmin[:, leaf], rmax[:, leaf]])
        node = leaf
        while node != NONE:
            box = np.concatenate([cmin[:, node], cmax[:, node]])
            for tmax in (float(wt[k]), 1.5 * float(wt[k]), float("inf")):
                if tmax <= 0.001:
                    continue
                if margin_leaf:
                    # the hit point lies strictly inside the leaf's cull box, hence inside every ancestor's
                    assert oracle.aabb_hit(box, org[k], d[k], 0.001, tmax), (name, int(k), int(node), tmax)
                elif oracle.aabb_hit(leaf_ref, org[k], d[k], 0.001, tmax):
                    # a leaf that kept the reference's box (which need not contain the surface): whenever the reference's own
                    # test of that box passes, so must the test of every cull box above it
                    assert oracle.aabb_hit(box, org[k], d[k], 0.001, tmax), (name, int(k), int(node), tmax)
                    implied += 1
            node = int(parent[node])
            checked += 1
    assert checked > 2000
    if name == "mixed_soup":
        assert implied > 100
