#!/usr/bin/env python3
"""Drives tools/layout_sim.cpp: line fetches per ray of the closest-hit walk over a 10^6-surface tree for several orders
of the node records in memory.  Host only (uses the CPU oracle's builder for the tree).

    python3 tools/layout_sim.py [spheres|triangles] [n_surfaces]
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from raytracing_amd import scenes as S  # noqa: E402

NONE = 0xFFFFFFFF
SIM_SURFACE = np.dtype([("kind", "<u4"), ("q", "<f8", (3,)), ("u", "<f8", (3,)), ("v", "<f8", (3,)), ("radius", "<f8")], align=True)


def treelet_layout(right, leaf, shape, include_root, recs_per_line=4):
    """Positions of the nodes when the tree is cut into treelets of `shape` (paths from the treelet's root: '' the root
    itself, 'L' / 'R' its children, ...) laid out one after the other in pre-order of their roots; a treelet never
    straddles a line if it fits in one."""
    n = len(right)
    pos = np.full(n, NONE, np.uint32)
    nxt = 0

    def child(i, c):
        if leaf[i] != NONE:
            return None
        return i + 1 if c == "L" else int(right[i])

    def place(members):
        nonlocal nxt
        room = recs_per_line - nxt % recs_per_line
        if room < len(members) <= recs_per_line:
            nxt += room
        for m in members:
            pos[m] = nxt
            nxt += 1

    stack = [0]
    if not include_root and n:
        place([0])
    while stack:
        x = stack.pop()
        members = []
        for path in shape:
            node = x
            for c in path:
                node = child(node, c) if node is not None else None
            if node is not None and pos[node] == NONE:
                members.append(node)
        place(members)
        # anchors of the next treelets, pushed so that they pop in pre-order
        anchors = []
        for m in members + [x]:
            if leaf[m] != NONE:
                continue
            kids = (m + 1, int(right[m]))
            if include_root:
                anchors.extend(c for c in kids if pos[c] == NONE)
            elif pos[kids[0]] == NONE:
                anchors.append(m)
        for a in sorted(set(anchors), reverse=True):
            stack.append(a)
    assert (pos != NONE).all()
    return pos, nxt


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "spheres"
    n_surf = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    lib_path = "/tmp/liblayout_sim.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", lib_path, os.path.join(ROOT, "tools", "layout_sim.cpp")])
    lib = C.CDLL(lib_path)
    lib.sim_visits.restype = C.c_uint64
    lib.sim_fetches.restype = C.c_uint64

    cache = f"/tmp/layout_tree_{kind}_{n_surf}.npz"
    surfaces, cam = S.synthetic_soup(n=n_surf, kind=kind, image_width=160)
    if os.path.exists(cache):
        z = np.load(cache)
        bmin, bmax, right, leaf = z["bmin"], z["bmax"], z["right"], z["leaf"]
    else:
        t0 = time.time()
        sc = po.Scene(surfaces, po.SAH_PER_SURFACE)
        bmin, bmax, right, leaf = sc.node_tables()
        np.savez(cache, bmin=bmin, bmax=bmax, right=right, leaf=leaf)
        print(f"built in {time.time() - t0:.1f} s", flush=True)
    n = len(right)
    ss = np.zeros(len(surfaces), SIM_SURFACE)
    for f in ("q", "u", "v", "radius"):
        ss[f] = surfaces[f]
    ss["kind"] = surfaces["geom_kind"]

    icam = po.camera_initialize(cam)
    org, d = po.primary_rays(icam, 1, 0)
    rng = np.random.default_rng(3)
    # secondary rays: from points inside the scene volume in random directions (what diffuse bounces look like)
    k = len(org)
    org2 = rng.uniform(-50, 50, (k, 3))
    d2 = rng.normal(size=(k, 3))
    org, d = np.concatenate([org, org2]), np.concatenate([d, d2])
    n_rays = len(org)

    def p(a, t):
        return a.ctypes.data_as(C.POINTER(t))

    bmin, bmax = np.ascontiguousarray(bmin), np.ascontiguousarray(bmax)
    right, leaf = np.ascontiguousarray(right, np.uint32), np.ascontiguousarray(leaf, np.uint32)
    offsets = np.zeros(n_rays + 1, np.uint64)
    args = (C.c_uint32(n), p(bmin, C.c_double), p(bmax, C.c_double), p(right, C.c_uint32), p(leaf, C.c_uint32), ss.ctypes.data_as(C.c_void_p),
            C.c_uint64(n_rays), p(org, C.c_double), p(d, C.c_double))
    total = lib.sim_visits(*args, None, None, C.c_uint64(0))
    seq = np.zeros(total, np.uint32)
    lib.sim_visits(*args, p(seq, C.c_uint32), p(offsets, C.c_uint64), C.c_uint64(total))
    print(f"{kind}: {n} nodes, {n_rays} rays, {total / n_rays:.1f} visits per ray", flush=True)

    # depth of every node
    depth = np.zeros(n, np.uint8)
    stack = [(0, 0)]
    while stack:
        i, dep = stack.pop()
        depth[i] = min(dep, 255)
        if leaf[i] == NONE:
            stack.append((int(right[i]), dep + 1))
            stack.append((i + 1, dep + 1))

    layouts = {"pre-order (implicit left child)": (np.arange(n, dtype=np.uint32), n)}
    for name, shape, incl in [
        ("left spine x4 {X,L,LL,LLL}", ["", "L", "LL", "LLL"], True),
        ("{X,L,LL,LR}", ["", "L", "LL", "LR"], True),
        ("{X,L,R}", ["", "L", "R"], True),
        ("{X,L,R,LL}", ["", "L", "R", "LL"], True),
        ("{X,L,R,RL}", ["", "L", "R", "RL"], True),
        ("sibling pairs {L,R}", ["L", "R"], False),
        ("pairs {L,R,LL,LR}", ["L", "R", "LL", "LR"], False),
        ("pairs {L,R,RL,RR}", ["L", "R", "RL", "RR"], False),
        ("7-node treelets {X,L,R,LL,LR,RL,RR}", ["", "L", "R", "LL", "LR", "RL", "RR"], True),
    ]:
        t0 = time.time()
        layouts[name] = treelet_layout(right, leaf, shape, incl)
        print(f"  layout {name}: {layouts[name][1] / n:.3f} records per node ({time.time() - t0:.1f} s)", flush=True)

    for top in (12, 14):
        print(f"top {top} levels always cached; line fetches per ray with the lane's last K lines cached")
        for name, (pos, used) in layouts.items():
            row = []
            for K in (1, 2, 3, 4):
                f = lib.sim_fetches(C.c_uint64(n_rays), p(seq, C.c_uint32), p(offsets, C.c_uint64), p(np.ascontiguousarray(pos), C.c_uint32),
                                    p(depth, C.c_uint8), C.c_uint32(top), C.c_uint32(K), C.c_uint32(4))
                row.append(f / n_rays)
            print(f"  {name:42s} " + "  ".join(f"K={K}: {v:7.2f}" for K, v in zip((1, 2, 3, 4), row)), flush=True)


if __name__ == "__main__":
    main()
