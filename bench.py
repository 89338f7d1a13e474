#!/usr/bin/env python3
"""Benchmark: Mrays/s (all bounces) and wall time of one frame.  The headline
is cover_spheres at 1920x1080, 500 samples per pixel; `--config` selects any of
the other benchmark configurations with the same line shape.

One step = one full render of the frame.  Inputs (the flattened BVH) are
resident in HBM before the timed region; the step covers clearing the
accumulation buffer, the whole sample loop, the cross-GPU sum (N > 1) and the
1/spp scaling, and ends with the finished linear image in HBM of rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]

With --gpus N > 1 and no WORLD_SIZE in the environment this process starts N
ranks of itself, one per GPU, before it touches a GPU, and prints rank 0's
line.  The launcher form works as well:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Samples per pixel are split over the ranks (500 -> 63,63,63,63,62,62,62,62 at
N = 8); sample s of pixel p uses RNG stream (seed, p, s) whatever the split,
and the per-rank f64 accumulation buffers are summed on rank 0 with one RCCL
reduce.  The job is the same frame at every N, so scaling is "strong".
`--single-process` renders the N-GPU frame from ONE process through the C
ABI's rtx_multi_render (one host thread per device, ncclCommInitAll, one
ncclReduce) instead of one process per GPU.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver supports dmabuf IPC only: without this RCCL cannot share buffers between the ranks of a node
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SEED, SCENE_SEED = 0xC0FFEE, 0xC0FFEE
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (spec)
FP64_VECTOR_PEAK_TF = 78.6   # MI355X vector FP64 peak (spec), 2 flop per FMA lane
KERNEL = "k_render_persistent"     # the single launch with its paths in a pool ...
KERNEL_RESIDENT = "k_render_resident"   # ... or in registers (scenes that sit in LDS whole, where paths bounce)


def frame_kernel(names):
    """Of the kernel names of a profile, the one that rendered the timed frames: the resident form if it ran at all (a scene's
    first call takes the pooled form), else the pooled form's non-counting instance."""
    names = list(names)
    for want in (KERNEL_RESIDENT, KERNEL + "<false"):
        hit = [k for k in names if k.replace("void ", "").replace("rtx::", "").startswith(want)]
        if hit:
            return hit[0]
    return None

# name -> (scene function kwargs -> (surfaces, camera), nominal spp, spp run by default, BVH built on: "host" | "device")
# The 10^6-surface frames take minutes at their nominal 1000 spp: they run at 20 spp unless --spp says otherwise
# (cost is linear in spp) and the metric string names the count that ran.
CONFIGS = {
    "cover_spheres": dict(spp=500, run_spp=500, build="host", width=1920),
    "hello_triangle": dict(spp=1, run_spp=1, build="host", width=1920),
    "cornell_box": dict(spp=20000, run_spp=20000, build="host", width=600),
    "quads": dict(spp=100, run_spp=100, build="host", width=400),
    "simple_light": dict(spp=1000, run_spp=1000, build="host", width=400),
    "simple_scene": dict(spp=100, run_spp=100, build="host", width=400),
    "demo_spheres": dict(spp=100, run_spp=100, build="host", width=400),
    "synthetic_1M_spheres": dict(spp=1000, run_spp=20, build="device", width=3840),
    "synthetic_1M_triangles": dict(spp=1000, run_spp=20, build="device", width=3840),
}


def make_scene(name, width, spp):
    from raytracing_amd import scenes as S

    if name == "cover_spheres":
        return S.cover_spheres(seed=SCENE_SEED, image_width=width, samples_per_pixel=spp)
    if name.startswith("synthetic_1M_"):
        return S.synthetic_soup(n=1_000_000, kind=name.split("_")[-1], image_width=width, samples_per_pixel=spp)
    surfaces, cam = S.EXAMPLES[name]()
    return surfaces, dict(cam, image_width=width, samples_per_pixel=spp)


def host_cpu():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    return model, os.cpu_count() or 1, affinity


def cpu_baseline(name, surfaces, cam, spp_nominal, target_s=12.0):
    """The CPU oracle (a C++ restatement of the reference renderer: same algorithm, f64, unordered stack traversal,
    six divisions per box test, rejection samplers) timed on this host, on every core the process may run on.
    A full frame is far too slow to time, so a bounded sample is rendered: the full frame at a few samples per
    pixel (cost is exactly linear in spp), or a strip of sample indices of a 1-spp frame for the largest scenes."""
    from oracle import pyoracle as po

    model, n_cpus, affinity = host_cpu()
    # Every core this process is allowed on.  OMP_NUM_THREADS, where the host sets it, states the share of the machine this job
    # was given; the run below uses the larger of the two so that no core the job may use is left idle, and says what it used.
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(affinity, 256))
    sc = po.Scene(surfaces, po.SAH_PER_SURFACE)
    icam = po.camera_initialize(cam)
    t0 = time.perf_counter()
    _, ctr = sc.render(icam, SEED, 0, 1, n_threads=threads)
    dt1 = time.perf_counter() - t0
    spp = max(1, min(64, int(target_s / max(dt1, 1e-3))))
    t0 = time.perf_counter()
    _, ctr = sc.render(icam, SEED, 1, spp, n_threads=threads)
    dt = time.perf_counter() - t0
    mrays = ctr.rays / dt / 1e6
    out = {
        "value": round(mrays, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "cores_available": n_cpus,
        "cores_in_affinity_mask": affinity,
        "host_thread_share_env": share or None,
        "cpu_model": model,
        "kind": "port",
        "sample": f"{name} {icam.image_width}x{icam.image_height} @ {spp} spp ({ctr.rays} rays in {dt:.2f} s on {threads} threads); "
                  f"the full {spp_nominal}-spp frame extrapolates to {dt * spp_nominal / spp:.0f} s",
        "wall_s_full_frame_extrapolated": round(dt * spp_nominal / spp, 1),
        "nodes_per_ray": round(ctr.node_visits / ctr.rays, 3),
        "sphere_tests_per_ray": round(ctr.sphere_tests / ctr.rays, 3),
        "planar_tests_per_ray": round(ctr.planar_tests / ctr.rays, 3),
    }
    if share and share != threads:
        # the same sample on exactly the job's stated share of the machine, for comparison
        t0 = time.perf_counter()
        _, c2 = sc.render(icam, SEED, 1, max(1, spp // 2), n_threads=share)
        d2 = time.perf_counter() - t0
        out["at_host_thread_share"] = {"threads": share, "value": round(c2.rays / d2 / 1e6, 3), "unit": "Mrays/s"}
        if out["at_host_thread_share"]["value"] > out["value"]:   # report the faster of the two as the baseline
            out["value"], out["cores"] = out["at_host_thread_share"]["value"], share
            out["wall_s_full_frame_extrapolated"] = round(d2 * spp_nominal / max(1, spp // 2), 1)
    sc.close()
    return out


def committed_traffic(name):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (fallback when the live passes cannot run)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        if name != "cover_spheres":
            rec = doc.get("configs", {}).get(name)
            return (rec["hbm_bytes_per_launch"], rec["rays_per_launch"], "profiles/hbm_traffic.json (committed): " + rec.get("source", "")) if rec else (None, None, None)
        recs = doc["kernels"]
        k = frame_kernel(recs)
        return recs[k]["hbm_bytes_per_launch"], recs[k].get("rays_per_launch"), "profiles/hbm_traffic.json (committed): " + recs[k].get("source", "")
    except (OSError, KeyError, ValueError, TypeError):
        return None, None, None


def committed_valu_issue(name):
    """How busy the vector ALUs were on the headline frame, from the counter passes committed under profiles/ (not measured
    in this run): vector instructions by kind x their issue cost (profiles/r01_valu_rate.txt: f64 and 64-bit integer
    arithmetic 4 cycles of a SIMD per wave instruction, f32 and 32-bit integer 2.2, f64 division / square root steps 16) over
    SIMDs x kernel cycles, and the share of lanes active in them.  The f64-flop fraction cannot say this: most vector
    instructions of the path are not f64 arithmetic (f32 box tests, the generator's integer rounds, selects, addresses)."""
    if name != "cover_spheres":
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r02_pmc_summary.json")) as fh:
            ks = json.load(fh)["kernels"]
        r = ks[frame_kernel(ks)]
        simd_cycles = 1024.0 * r["GRBM_GUI_ACTIVE"] / 8.0   # the counter is summed over the eight XCDs
        f64 = r["SQ_INSTS_VALU_ADD_F64"] + r["SQ_INSTS_VALU_FMA_F64"] + r["SQ_INSTS_VALU_MUL_F64"]
        slow = f64 + r["SQ_INSTS_VALU_INT64"] + r["SQ_INSTS_VALU_CVT"]
        fast = r["SQ_INSTS_VALU"] - slow - r["SQ_INSTS_VALU_TRANS_F64"]
        return {"frac": round((4.0 * slow + 16.0 * r["SQ_INSTS_VALU_TRANS_F64"] + 2.2 * fast) / simd_cycles, 3),
                "frac_if_every_instruction_took_4_cycles": round(4.0 * r["SQ_INSTS_VALU"] / simd_cycles, 3),
                "lanes_active": round(r["SQ_THREAD_CYCLES_VALU"] / (r["SQ_ACTIVE_INST_VALU"] * 64.0), 3),
                "source": "profiles/r02_pmc_summary.json (committed, not of this run), issue costs from profiles/r01_valu_rate.txt"}
    except (OSError, KeyError, ValueError, StopIteration, ZeroDivisionError, TypeError):
        return None


def live_traffic(args):
    """HBM bytes of one launch of the render kernel on THIS box: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do
    not fit in one) over a child run of this script that renders one frame.  FETCH_SIZE counts a 128-byte request as 64 bytes
    on gfx950, so reads are doubled; both counters are in KiB.  Returns (bytes, rays of that launch, source) or Nones."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="rtx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    # one warm-up frame as in the timed run: a scene's first call always takes the pooled form of the kernel
    child = ["python3", os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", args.config, "--spp", str(args.spp), "--width", str(args.width),
             "--steps", "1", "--warmup", "1"]
    got, rays = {}, None
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["timeout", "-s", "KILL", "240", rocprof, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", *child]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd="/tmp", env=env)
            if r.returncode != 0:
                return None, None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode})"
            for line in r.stdout.splitlines():
                if line.startswith("{"):
                    rays = json.loads(line)["config"]["rays_per_frame"]
            per_kernel = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == ctr:
                            tot = per_kernel.setdefault(row["Kernel_Name"], [0.0, set()])
                            tot[0] += float(row["Counter_Value"])
                            tot[1].add(row["Dispatch_Id"])
            k = frame_kernel(per_kernel)
            if k is None:
                return None, None, f"no {KERNEL} / {KERNEL_RESIDENT} dispatch in the {ctr} pass"
            kernel_seen = k.replace("void ", "").replace("rtx::", "").split("(")[0]
            got[ctr] = per_kernel[k][0] * 1024.0 / len(per_kernel[k][1])
        return 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], rays, (
            f"live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one pass each) of {kernel_seen} over `python3 bench.py --pmc-child` on this box; "
            f"read {2.0 * got['FETCH_SIZE'] / 1e9:.1f} GB (FETCH_SIZE x 2) + written {got['WRITE_SIZE'] / 1e9:.1f} GB")
    except (OSError, ValueError, KeyError) as e:
        return None, None, f"live PMC passes failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, json_out):
    """`python bench.py --gpus N` typed as is: N fresh ranks of this script, one per GPU.  Runs before this process has
    made any GPU call (it never makes one); prints what rank 0 printed and returns the worst exit status."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *[a for a in sys.argv[1:] if a != "--force-spawn"]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, abs(p.wait()))
    json_out.write(out)
    json_out.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cover_spheres", choices=sorted(CONFIGS))
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (default: the configuration's own)")
    ap.add_argument("--width", type=int, default=0, help="image width (default: the configuration's own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 counter passes behind roofline.traffic")
    ap.add_argument("--single-process", action="store_true", help="N GPUs from one process through rtx_multi_render")
    ap.add_argument("--time-kernels", action="store_true", help="round-loop path with per-kernel HIP event timing (diagnostic)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # one bare frame under rocprofv3
    ap.add_argument("--force-spawn", action="store_true", help="start the ranks as child processes even for --gpus 1 (exercises the launcher)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    # Libraries below (RCCL prints a version banner) may write to stdout; the one JSON line must be all that this process
    # writes there: everything else goes to stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    args.spp = args.spp or cfg["run_spp"]
    args.width = args.width or cfg["width"]

    if (args.gpus > 1 or args.force_spawn) and "WORLD_SIZE" not in os.environ and not args.single_process:
        raise SystemExit(spawn_ranks(args.gpus, json_out))

    import torch
    import torch.distributed as dist

    import raytracing_amd as rt
    from raytracing_amd.multi_gpu import collective_ranks, reduce_accum, spp_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = 1 if args.single_process else int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.single_process:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if "WORLD_SIZE" in os.environ and not args.single_process:   # also a world of one, so that the launcher path is the same at every N
        dist.init_process_group("nccl", device_id=dev)
        ranks_seen = collective_ranks(dev)   # also sets the communicator up before anything is timed

    # host side: scene, BVH (SAH PerSurface, the reference's default), camera
    surfaces, cam = make_scene(args.config, args.width, args.spp)
    t0 = time.perf_counter()
    bvh = rt.HostBVH(surfaces, rt.SAH_PER_SURFACE, device=cfg["build"] == "device")
    bvh_ms = (time.perf_counter() - t0) * 1e3
    icam = rt.camera_initialize(cam)
    W, H = icam.image_width, icam.image_height
    nval = W * H * 3
    stream = torch.cuda.current_stream().cuda_stream

    if args.single_process:
        multi = rt.MultiScene(bvh, args.gpus)   # scene on every device + ncclCommInitAll: outside the timed region
        ranks_seen = None

        def step(flags=0):
            nonlocal ranks_seen
            _, ms = multi.render(icam, SEED, flags, to_host=False)
            ranks_seen = ms.collective_ranks
            st = rt.Stats()
            st.rays, st.paths, st.ms_total = ms.rays, ms.paths, max(ms.ms_device[: args.gpus])
            return st
        scene = None
        begin, count = 0, args.spp
    else:
        scene = rt.Scene(bvh)   # BVH now resident in HBM
        begin, count = spp_range(rank, world, args.spp)
        acc = torch.zeros(H, W, 3, dtype=torch.float64, device=dev)
        image = torch.empty_like(acc)

        def step(flags=0):
            acc.zero_()
            st = scene.render_accumulate(icam, SEED, begin, count, acc.data_ptr(), stream, flags) if count else rt.Stats()
            reduce_accum(acc, world)
            if rank == 0:
                rt.scale(acc.data_ptr(), nval, icam.pixel_samples_scale, image.data_ptr(), stream)
            return st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:   # the reduce's channels for a buffer of this size, before anything is timed (also with --warmup 0)
        reduce_accum(acc, world)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stats = [step(rt.FLAG_TIME_KERNELS if args.time_kernels else 0) for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    work = torch.tensor([sum(s.rays for s in stats), sum(s.paths for s in stats)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays_per_step = work[0].item() / args.steps
    paths_per_step = work[1].item() / args.steps

    if rank == 0:
        n_gpus = args.gpus
        ms_per_step = elapsed * 1e3 / args.steps
        mrays = rays_per_step / (ms_per_step * 1e3)
        nominal = (args.spp, args.width) == (cfg["spp"], cfg["width"])
        what = f"{args.config} {W}x{H} @{args.spp} spp" + ("" if args.spp == cfg["spp"] else f" (the configuration's frame has {cfg['spp']})")
        out = {
            "metric": f"Mrays/s (all bounces), {what}",
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "wall_s_per_frame": round(ms_per_step / 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (seeded scene built by the example's procedure)" if args.config in ("cover_spheres",) or args.config.startswith("synthetic")
                    else "the example's scene as data",
            "config": {
                "workload": f"{what}, max_depth {icam.max_depth}, {len(surfaces)} surfaces, {bvh.n_nodes} BVH nodes (SAH PerSurface, built on the {cfg['build']})",
                "nominal": nominal,
                "parallelism": ("single GPU" if n_gpus == 1 else
                                f"spp split over {n_gpus} GPUs, one RCCL reduce; " + ("one process, one host thread per GPU (rtx_multi_render)" if args.single_process
                                                                                     else "one process per GPU (torch.distributed, nccl backend)")),
                "ranks_in_collective": ranks_seen,
                "rays_per_frame": int(rays_per_step),
                "paths_per_frame": int(paths_per_step),
                "bvh_build_ms": round(bvh_ms, 2),
            },
        }
        if args.pmc_child or scene is None:   # a bare frame for the counter passes / the single-process form: no counting pass
            print(json.dumps(out), file=json_out, flush=True)
        else:
            out["roofline"] = roofline(rt, scene, bvh.n_nodes, icam, stats, begin, count, acc, stream, args)
            if args.time_kernels:   # round-loop path: summed launch durations per kernel
                out["roofline"]["kernel_time_share"] = {"traverse_ms": round(sum(s.ms_traverse for s in stats) / args.steps, 2),
                                                        "shade_ms": round(sum(s.ms_shade for s in stats) / args.steps, 2),
                                                        "generate_ms": round(sum(s.ms_generate for s in stats) / args.steps, 2)}
            if not args.no_cpu_baseline and world == 1:
                out["cpu_baseline"] = cpu_baseline(args.config, surfaces, cam, cfg["spp"])
                out["speedup_vs_cpu_baseline"] = round(mrays / out["cpu_baseline"]["value"], 1)
            print(json.dumps(out), file=json_out, flush=True)

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def roofline(rt, scene, n_nodes, icam, stats, begin, count, acc, stream, args):
    """Roofline of the one kernel a frame is (k_render_persistent, one launch per step), from this rank's launches.

    Work per ray is algorithmic: Nv box tests, Ns sphere tests, Np planar tests per ray from a counting pass of the same
    kernel (equal to the CPU oracle's counters), priced at
        flops = 37 Nv + 30 Ns + 57 Np + 60
        bytes = 56 Nv + 32 Ns + 128 Np (nodes, spheres, planars) + 40 (material) + 160 (ray state, read + write) + 12 (hit record)
                + 12 per path (accumulation).
    Launch time is the HIP-event time of the launch on its stream (rtx_stats.ms_total).
    Which bound applies depends on where the tree lives.  Scenes whose node records sit in LDS (every example scene)
    are bound by f64 vector arithmetic, latency and divergence: `frac` is the f64-VALU fraction, and only the ray state, hit
    records, material records and accumulation of the byte formula can reach HBM (`hbm.algorithmic`).  The 10^6-surface scenes
    walk their tree from global memory: `frac` is the byte formula with the node record this design reads (16 or 32 B per
    visit instead of the 56 B of an SoA f64 node: 48 B box + two links) over the HBM peak; the counters see more (`hbm.measured_over_algorithmic`): every
    16-byte record read that misses L2 moves a whole line."""
    cnt = scene.render_accumulate(icam, SEED, begin, min(count, 2), acc.data_ptr(), stream, rt.FLAG_COUNT_WORK)
    nv, ns, npl = cnt.node_visits / cnt.rays, cnt.sphere_tests / cnt.rays, cnt.planar_tests / cnt.rays
    rank_rays = sum(s.rays for s in stats)
    rank_paths = sum(s.paths for s in stats)
    launches = len(stats)
    rays_per_launch = rank_rays / launches
    avg_launch_s = sum(s.ms_total for s in stats) / launches * 1e-3
    path_share = 12.0 * rank_paths / max(rank_rays, 1)
    scene_bytes_per_ray = 56.0 * nv + 32.0 * ns + 128.0 * npl
    state_bytes_per_ray = 212.0 + path_share
    flops_per_ray = 37.0 * nv + 30.0 * ns + 57.0 * npl + 60.0
    tf = flops_per_ray * rays_per_launch / avg_launch_s / 1e12
    gbs_state = state_bytes_per_ray * rays_per_launch / avg_launch_s / 1e9
    gbs_all = (scene_bytes_per_ray + state_bytes_per_ray) * rays_per_launch / avg_launch_s / 1e9
    tree_on_chip = n_nodes * 32 <= 150 * 1024   # the single launch keeps trees up to 150 KB of f32 records in LDS
    if not tree_on_chip:
        # What the walk reads per node from global memory: 16-byte quantised records for trees beyond 4 MiB of f32 records
        # (RTX_NODE16 overrides), else the 32-byte f32 records; the 56 bytes of an SoA f64 node (48 B box + two links) are what nobody reads.
        rec = 16.0 if int(os.environ.get("RTX_NODE16", "1" if n_nodes * 32 > (4 << 20) else "0")) else 32.0
        soa_f64_gbs = gbs_all
        scene_bytes_per_ray = rec * nv + 32.0 * ns + 128.0 * npl
        gbs_all = (scene_bytes_per_ray + state_bytes_per_ray) * rays_per_launch / avg_launch_s / 1e9

    traffic = t_rays = src = None
    if not args.no_pmc and args.gpus == 1:
        traffic, t_rays, src = live_traffic(args)
    if traffic is None:
        live_note = src
        traffic, t_rays, src = committed_traffic(args.config)
        if live_note and src:
            src += f" [{live_note}]"
    if traffic is not None and t_rays:   # the counted launch may have traced a few more or fewer rays (another seed or spp): scale per ray
        traffic = traffic * rays_per_launch / t_rays
    hbm = {
        "peak_GBps": HBM_PEAK_GBS,
        "algorithmic_bytes_per_ray": round(state_bytes_per_ray if tree_on_chip else scene_bytes_per_ray + state_bytes_per_ray, 1),
        "algorithmic_GBps": round(gbs_state if tree_on_chip else gbs_all, 1),
        "algorithmic_frac": round((gbs_state if tree_on_chip else gbs_all) / HBM_PEAK_GBS, 4),
        "measured_bytes_per_ray": round(traffic / rays_per_launch, 1) if traffic else None,
        "measured_GBps": round(traffic / avg_launch_s / 1e9, 1) if traffic else None,
        "measured_frac": round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
        "measured_over_algorithmic": round(traffic / rays_per_launch / (state_bytes_per_ray if tree_on_chip else scene_bytes_per_ray + state_bytes_per_ray), 3) if traffic else None,
    }
    common = {
        "kernel": KERNEL_RESIDENT if KERNEL_RESIDENT in (src or "") else KERNEL,
        "traffic": round(traffic, 1) if traffic else None,
        "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE)",
        "traffic_source": src,
        "rays_per_launch": round(rays_per_launch, 1),
        "avg_launch_ms": round(avg_launch_s * 1e3, 4),
        "launches": int(launches),
        "nodes_per_ray": round(nv, 3),
        "sphere_tests_per_ray": round(ns, 3),
        "planar_tests_per_ray": round(npl, 3),
        "flops_per_ray": round(flops_per_ray, 1),
        "box_tests_left_to_f64_tiers": cnt.box_fallbacks / max(cnt.node_visits, 1),
        "rounds_of_busiest_wave": int(max(s.iterations for s in stats)),
        "hbm": hbm,
        "valu_f64": {"achieved_tflops": round(tf, 3), "peak_tflops": FP64_VECTOR_PEAK_TF, "frac": round(tf / FP64_VECTOR_PEAK_TF, 4)},
        "valu_issue": committed_valu_issue(args.config),
        "on_chip_note": f"{scene_bytes_per_ray:.0f} B/ray of the byte formula are node and surface records "
                        + ("served from LDS (the tree and the surface tables are copied there once per block): not HBM traffic"
                           if tree_on_chip else "read from global memory (the tree is far larger than LDS and L2)")
                        + ("; in the resident form of the kernel the paths stay in registers from bounce to bounce, so the formula's ray state and hit "
                           "records do not move either: what is measured is the image's atomic adds" if KERNEL_RESIDENT in (src or "") else ""),
    }
    if tree_on_chip:
        return dict({"bound": "valu_f64", "achieved": round(tf, 3), "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(tf / FP64_VECTOR_PEAK_TF, 4)}, **common)
    common["soa_f64_node_formula"] = {"bytes_per_ray": round(56.0 * nv + 32.0 * ns + 128.0 * npl + state_bytes_per_ray, 1), "GBps": round(soa_f64_gbs, 1),
                                "note": "56 B per node visit (an SoA f64 node): more than this design reads per visit, so not a rate the memory system sees"}
    common["node_record_bytes"] = rec
    return dict({"bound": "hbm", "achieved": round(gbs_all, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs_all / HBM_PEAK_GBS, 4)}, **common)


if __name__ == "__main__":
    main()
