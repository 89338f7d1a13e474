"""rtx_multi_*: the N-GPU frame from one process (one host thread per device, ncclCommInitAll, one ncclReduce),
through the C ABI.  A one-GPU box can only run the world of one; that still goes through the communicator set-up,
the reduce and the scaling, and must give the single-GPU image."""
import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu


def test_multi_render_on_the_visible_devices_equals_single_device_render(rt):
    surfaces, cam = S.cover_spheres()
    cam = dict(cam, image_width=256, samples_per_pixel=9)
    bvh = rt.HostBVH(surfaces)
    icam = rt.camera_initialize(cam)
    want, st = rt.Scene(bvh).render(icam, 17)
    for n in sorted({1, min(rt.device_count(), 2), rt.device_count()}):
        multi = rt.MultiScene(bvh, n)
        for _ in range(2):   # the handle is reusable; the second frame starts from cleared buffers
            img, ms = multi.render(icam, 17)
            assert (ms.n_gpus, ms.collective_ranks) == (n, n)
            assert (ms.rays, ms.paths) == (st.rays, st.paths)
            assert list(ms.spp_count[:n]) == [9 // n + (1 if g < 9 % n else 0) for g in range(n)]
            assert sum(ms.spp_count[:n]) == 9 and ms.spp_begin[0] == 0
            np.testing.assert_allclose(img, want, rtol=1e-13, atol=0)   # same radiance values, summed in another order
            assert ms.ms_frame > 0 and ms.ms_frame >= max(ms.ms_device[:n])
        assert multi.image_ptr()
        multi.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_more_ranks_than_devices_rehearsed_from_one_process(rt, monkeypatch, world):
    """RTX_MULTI_SHARE_GPU=1: rtx_multi_* with more ranks than there are devices.  Every rank keeps its own scene handle,
    stream, host thread and share of the samples; the ranks share devices, so the sum onto rank 0 is made with copies and
    adds instead of the collective (RCCL refuses two ranks on one device).  What it checks is everything but the
    collective: the sample ranges in C++, concurrent frames on separate handles, the totals, the scaling, the caller's device."""
    import torch

    if world <= rt.device_count():
        pytest.skip("this many devices are really there")
    monkeypatch.setenv("RTX_MULTI_SHARE_GPU", "1")
    surfaces, cam = S.cover_spheres()
    cam = dict(cam, image_width=200, samples_per_pixel=11)
    bvh = rt.HostBVH(surfaces)
    icam = rt.camera_initialize(cam)
    want, st = rt.Scene(bvh).render(icam, 23)
    before = torch.cuda.current_device()
    multi = rt.MultiScene(bvh, world)
    for _ in range(2):
        img, ms = multi.render(icam, 23)
        assert (ms.n_gpus, ms.collective_ranks) == (world, world)
        assert (ms.rays, ms.paths) == (st.rays, st.paths)
        assert list(ms.spp_count[:world]) == [11 // world + (1 if g < 11 % world else 0) for g in range(world)]
        assert [ms.spp_begin[g] for g in range(world)] == [sum(ms.spp_count[:g]) for g in range(world)]
        np.testing.assert_allclose(img, want, rtol=1e-13, atol=0)
    multi.close()
    assert torch.cuda.current_device() == before
    monkeypatch.delenv("RTX_MULTI_SHARE_GPU")
    with pytest.raises(rt.RtxError):
        rt.MultiScene(bvh, rt.device_count() + 1)   # without the variable the count of devices is the limit, as before


def test_one_shot_render_multi_and_errors(rt):
    import ctypes as C

    surfaces, cam = S.simple_scene()
    bvh = rt.HostBVH(surfaces)
    icam = rt.camera_initialize(dict(cam, image_width=64, samples_per_pixel=3))
    want, _ = rt.Scene(bvh).render(icam, 2)
    img = np.empty_like(want)
    ms = rt.MultiStats()
    rc = rt._lib.rtx_render_multi(C.byref(bvh.desc), C.byref(icam), 2, 1, 0, img.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ms))
    assert rc == 0, rt._lib.rtx_last_error()
    np.testing.assert_allclose(img, want, rtol=1e-13, atol=0)
    assert ms.collective_ranks == 1
    for bad in (0, -1, rt.device_count() + 1, 1000):
        with pytest.raises(rt.RtxError) as e:
            rt.MultiScene(bvh, bad)
        assert e.value.code == -1


def test_runner_on_several_gpus_writes_the_same_image(rt, oracle, tmp_path):
    """RenderRunner with n_gpus set (rth_run_multi; bin/rtx_example --gpus N): image and P3 text as rth_run's, up to the
    order of the f64 additions (one device: the same values through the communicator, the reduce and the scaling)."""
    import ctypes as C

    surfaces, cam = S.EXAMPLES["simple_scene"]()
    cfg = rt.camera_config(dict(cam, samples_per_pixel=5, image_width=120))
    icam = rt.camera_initialize(cfg)
    single = np.zeros((icam.image_height, icam.image_width, 3))
    multi = np.zeros_like(single)
    ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert rt._lib.rth_run(surfaces.ctypes.data_as(C.c_void_p), len(surfaces), 3, 0, C.byref(cfg), 21, None, ptr(single)) == 0
    n = min(rt.device_count(), 2)
    path = tmp_path / "multi.ppm"
    rc = rt._lib.rth_run_multi(surfaces.ctypes.data_as(C.c_void_p), len(surfaces), 3, 0, C.byref(cfg), 21, n, str(path).encode(), ptr(multi))
    assert rc == 0, rt._lib.rth_last_error()
    np.testing.assert_allclose(multi, single, rtol=1e-13, atol=0)
    q = oracle.ppm_quantize(multi.reshape(-1, 3))
    want = f"P3\n{icam.image_width} {icam.image_height}\n255\n" + "".join(f"{r} {g} {b}\n" for r, g, b in q.tolist())
    assert path.read_bytes() == want.encode()
    assert rt._lib.rth_run_multi(surfaces.ctypes.data_as(C.c_void_p), len(surfaces), 3, 0, C.byref(cfg), 21, 0, None, ptr(multi)) == -1


@pytest.mark.parametrize("how", ["--force-spawn", "--single-process"])
def test_bench_launch_paths_with_a_world_of_one(how):
    """bench.py's two N-GPU launch paths with N = 1 (all a one-GPU box can run): ranks started as child processes of
    `python bench.py --gpus N` (watched, one per GPU, torch.distributed over RCCL), and the frame through rtx_multi_render
    from one process.  The line must say that the collective saw every rank."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), how, "--gpus", "1", "--steps", "1", "--warmup", "0", "--spp", "8", "--width", "480",
           "--no-cpu-baseline", "--no-pmc", "--no-cold-frame"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout          # stdout carries the JSON line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["ranks_in_collective"] == 1
    assert d["value"] > 0 and d["config"]["paths_per_frame"] == 480 * 270 * 8
    if how == "--single-process":
        assert d["per_rank"]["spp_count"] == [8] and d["per_rank"]["ms_render"][0] > 0 and d["per_rank"]["ms_reduce"] >= 0
    else:
        assert d["image_check"]["rays_equal"] and d["image_check"]["max_rel_diff"] <= 1e-12
        assert d["image_check"]["timed_form"] != d["image_check"]["other_form"]


def test_bench_with_two_ranks_rehearsed_on_the_gpus_there_are():
    """The N > 1 path of `python bench.py --gpus N` end to end on a box with fewer GPUs: two ranks, each a process of its own
    started and watched by the launcher, share the GPU (`--share-gpu`) and sum their buffers through host memory
    (`--backend gloo`: RCCL refuses two ranks on one device).  Everything but the collective's backend is what an 8-GPU run
    does: rendezvous, the split of the samples, the reduce onto rank 0, the per-rank fields, the maximum over ranks.  The
    frame's ray count must equal the single-rank run's: each (pixel, sample) has its own random stream, so it is the same set
    of paths only if the two ranks' ranges cover the samples exactly once."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "2", "--warmup", "1", "--spp", "9", "--width", "480", "--no-cpu-baseline", "--no-pmc", "--no-cold-frame", "--no-image-check"]
    lines = {}
    for n in (1, 2):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--force-spawn", "--backend", "gloo", "--share-gpu", *common]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(out) == 1, r.stdout
        lines[n] = json.loads(out[0])
    one, two = lines[1], lines[2]
    assert two["n_gpus"] == 2 and two["config"]["ranks_in_collective"] == 2
    assert two["per_rank"]["spp_count"] == [5, 4]
    assert all(v > 0 for v in two["per_rank"]["ms_render"]) and len(two["per_rank"]["ms_reduce"]) == 2
    assert two["config"]["paths_per_frame"] == one["config"]["paths_per_frame"] == 480 * 270 * 9
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
    assert "REHEARSAL" in two["config"]["parallelism"]
    a, b = one["config"]["image_checksum"], two["config"]["image_checksum"]
    assert a > 0 and abs(a - b) <= 1e-11 * a          # the summed image of two ranks is the image of one


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n, extra", [(1, []), (2, ["--backend", "gloo", "--share-gpu"])])
def test_bench_under_the_launcher_the_scaling_run_uses(n, extra):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    --steps K --warmup W`: the command an N-GPU scaling run is started with.  Under it bench.py is one of the ranks (it starts no
    ranks of its own), rank 0 prints the one JSON line and the others print nothing.  With one rank over RCCL, and with two ranks
    that share the GPU and sum through host memory (what a one-GPU box can run of N = 2)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--spp", "9", "--width", "480", "--no-cpu-baseline", "--no-pmc", "--no-cold-frame", "--no-image-check", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["ranks_in_collective"] == n
    assert d["config"]["paths_per_frame"] == 480 * 270 * 9 and d["value"] > 0
    if n == 2:
        assert d["per_rank"]["spp_count"] == [5, 4] and "REHEARSAL" in d["config"]["parallelism"]
