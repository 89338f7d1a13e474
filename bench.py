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


def short_kernel(name):
    return name.replace("void ", "").replace("rtx::", "").split("(")[0].strip()


def frame_kernel(names, want):
    """Of the kernel names of a profile, the instance of `want` (the kernel rtx_stats.kernel_form named for the timed frames)
    that rendered them: for the pooled form its non-counting instance."""
    prefix = want + "<false" if want == KERNEL else want
    hit = [k for k in names if short_kernel(k).startswith(prefix)]
    return hit[0] if hit else None

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


def committed_traffic(name, want):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (fallback when the live passes cannot run)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        if name != "cover_spheres":
            rec = doc.get("configs", {}).get(name)
            return (rec["hbm_bytes_per_launch"], rec["rays_per_launch"], "profiles/hbm_traffic.json (committed): " + rec.get("source", "")) if rec else (None, None, None)
        recs = doc["kernels"]
        k = frame_kernel(recs, want)
        return recs[k]["hbm_bytes_per_launch"], recs[k].get("rays_per_launch"), "profiles/hbm_traffic.json (committed): " + recs[k].get("source", "")
    except (OSError, KeyError, ValueError, TypeError):
        return None, None, None


# Counter passes of one frame, each a rocprofv3 run of its own (a run holds at most 8 SQ counters; FETCH_SIZE and WRITE_SIZE
# do not fit together).  The last two price how busy the vector ALUs were.
PMC_PASSES = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT64",
     "SQ_INSTS_VALU_CVT", "GRBM_GUI_ACTIVE"],
    ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_SALU"],
]


def valu_issue(c, rays):
    """How busy the vector ALUs were, from this run's counters of the frame's kernel (per launch): vector instructions by kind x
    their issue cost (profiles/r01_valu_rate.txt: f64 and 64-bit integer arithmetic 4 cycles of a SIMD per wave instruction,
    f32 and 32-bit integer 2.2, f64 division / square root steps 16) over SIMDs x kernel cycles, and the share of lanes active in
    them.  The f64-flop fraction cannot say this: most vector instructions of the path are not f64 arithmetic (f32 box tests,
    the generator's integer rounds, selects, addresses)."""
    try:
        simd_cycles = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0   # the counter is summed over the eight XCDs
        f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"]
        slow = f64 + c["SQ_INSTS_VALU_INT64"] + c["SQ_INSTS_VALU_CVT"]
        fast = c["SQ_INSTS_VALU"] - slow - c["SQ_INSTS_VALU_TRANS_F64"]
        return {"frac": round((4.0 * slow + 16.0 * c["SQ_INSTS_VALU_TRANS_F64"] + 2.2 * fast) / simd_cycles, 3),
                "frac_if_every_instruction_took_4_cycles": round(4.0 * c["SQ_INSTS_VALU"] / simd_cycles, 3),
                "lanes_active": round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0), 3),
                "wave_cycles_waiting": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                "wave_cycles_ready_but_not_issued": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                "vector_instructions_per_ray": round(c["SQ_INSTS_VALU"] / rays, 2) if rays else None,
                "scalar_instructions_per_ray": round(c["SQ_INSTS_SALU"] / rays, 2) if rays else None,
                "source": "live: rocprofv3 --pmc passes of this run (per launch of the frame's kernel), issue costs from profiles/r01_valu_rate.txt"}
    except (KeyError, ZeroDivisionError, TypeError):
        return None


def live_counters(args, want):
    """Counters of one launch of the kernel that renders the frame (`want`), on THIS box: one rocprofv3 --pmc run per entry of
    PMC_PASSES over a child run of this script that renders one frame after one warm-up frame, as the timed run does.  Returns
    (per-launch counter values, rays of that launch, the kernel's name as the profiler saw it, note)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {}, None, None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="rtx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    child = ["python3", os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", args.config, "--spp", str(args.spp), "--width", str(args.width),
             "--steps", "1", "--warmup", "1"]
    got, rays, seen, notes = {}, None, None, []
    try:
        for i, ctrs in enumerate(PMC_PASSES):
            d = os.path.join(tmp, f"pass{i}")
            cmd = ["timeout", "-s", "KILL", "240", rocprof, "--pmc", *ctrs, "--output-format", "csv", "-d", d, "--", *child]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd="/tmp", env=env)
            if r.returncode != 0:
                notes.append(f"rocprofv3 --pmc {' '.join(ctrs)} failed (rc {r.returncode})")
                continue
            for line in r.stdout.splitlines():
                if line.startswith("{"):
                    rays = json.loads(line)["config"]["rays_per_frame"]
            per_kernel = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        tot = per_kernel.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], [0.0, set()])
                        tot[0] += float(row["Counter_Value"])
                        tot[1].add(row["Dispatch_Id"])
            k = frame_kernel(per_kernel, want)
            if k is None:
                notes.append(f"no {want} dispatch in the pass of {' '.join(ctrs)}")
                continue
            seen = short_kernel(k)
            for c, (total, dispatches) in per_kernel[k].items():
                got[c] = total / len(dispatches)
    except (OSError, ValueError, KeyError) as e:
        notes.append(f"live PMC passes failed: {e}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return got, rays, seen, "; ".join(notes) or None


def cold_frame(args):
    """What a user of RenderRunner::run sees (src/runner.rs:24-38: ONE frame per process): a fresh process of bin/rtx_example
    renders the configuration's frame and prints the reference's three timers; RTX_LOG_COLD=1 makes the library add the
    host-side cost of what a process pays once (code object load, path pool, the first call's probe launch)."""
    exe = os.path.join(ROOT, "bin", "rtx_example")
    if args.config.startswith("synthetic") or not os.path.exists(exe):
        return None
    out = tempfile.NamedTemporaryFile(prefix="rtx_cold_", suffix=".ppm", dir="/tmp", delete=False)
    out.close()
    env = dict(os.environ, RTX_LOG_COLD="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RTX_RESIDENT"):
        env.pop(k, None)
    cmd = [exe, args.config, "--width", str(args.width), "--spp", str(args.spp), "--seed", str(SEED), "-o", out.name]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(["timeout", "-k", "10", "300", *cmd], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
    finally:
        wall = time.perf_counter() - t0
        try:
            os.unlink(out.name)
        except OSError:
            pass
    if r.returncode != 0:
        return {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
    res = {"command": "RTX_LOG_COLD=1 bin/rtx_example " + " ".join(cmd[1:-2]) + " -o <file>", "process_wall_s": round(wall, 4), "once_per_process_ms": {}}
    for line in r.stderr.splitlines():
        line = line.strip()
        for key, name in (("Total runtime:", "total_runtime_s"), ("BVH construction:", "bvh_construction_s"), ("Rendering:", "rendering_s")):
            if line.startswith(key):
                res[name] = float(line.split(":")[1].strip().rstrip("s"))
        if line.startswith("rays:") and "device:" in line:
            res["device_ms"] = float(line.split("device:")[1].split("ms")[0])
        if line.startswith("rtx cold:"):
            what, ms = line[len("rtx cold:"):].rsplit(" ", 2)[0].strip(), float(line.rsplit(" ", 2)[1])
            res["once_per_process_ms"][what] = round(res["once_per_process_ms"].get(what, 0.0) + ms, 3)
    res["note"] = ("the reference's timers: total = BVH construction + rendering; rendering spans scene upload, the render call (device_ms of it "
                   "on the GPU), the P3 text made on the device, its copy to the host and the write of the file")
    return res


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, json_out, grace_s=120.0):
    """`python bench.py --gpus N` typed as is: N fresh ranks of this script, one per GPU.  Runs before this process has
    made any GPU call (it never makes one).  All ranks are watched: when one exits with an error the others are ended (they
    would wait for it in a collective until some outer limit), its last lines of stderr are shown, and the status is non-zero
    within seconds; ranks that outlive the first clean exit by more than `grace_s` are ended as well.  Prints what rank 0
    printed and returns the worst exit status."""
    port = free_port()
    procs, logs = [], []
    argv = [a for a in sys.argv[1:] if a != "--force-spawn"]
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        log = tempfile.TemporaryFile(mode="w+", prefix=f"rtx_rank{r}_", dir="/tmp")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=log, text=True))

    def tail(r, lines=25):
        logs[r].seek(0)
        return "".join(logs[r].readlines()[-lines:])

    def end_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    rc, first_exit = 0, None
    while any(p.poll() is None for p in procs):
        failed = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if failed:
            r = failed[0]
            rc = abs(procs[r].returncode) or 1
            print(f"bench.py: rank {r} of {n} exited with status {procs[r].returncode}; ending the other ranks. Its last lines:\n{tail(r)}", file=sys.stderr, flush=True)
            end_all()
            break
        if first_exit is None and any(p.poll() == 0 for p in procs):
            first_exit = time.time()
        if first_exit is not None and time.time() - first_exit > grace_s:
            late = [r for r, p in enumerate(procs) if p.poll() is None]
            print(f"bench.py: ranks {late} still running {grace_s:.0f} s after the first rank finished; ending them. Rank {late[0]}'s last lines:\n{tail(late[0])}",
                  file=sys.stderr, flush=True)
            rc = 1
            end_all()
            break
        time.sleep(0.05)
    for r, p in enumerate(procs):
        if p.returncode not in (None, 0) and rc == 0:
            rc = abs(p.returncode) or 1
            print(f"bench.py: rank {r} of {n} exited with status {p.returncode}. Its last lines:\n{tail(r)}", file=sys.stderr, flush=True)
    out = procs[0].stdout.read() if procs[0].stdout else ""
    if rc == 0:
        sys.stderr.write(tail(0, 1000))   # what rank 0 had to say (RCCL's banner and the like)
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
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 counter passes behind roofline.traffic and roofline.valu_issue")
    ap.add_argument("--no-image-check", action="store_true", help="do not render the frame once more in the other form of the kernel")
    ap.add_argument("--no-cold-frame", action="store_true", help="do not time a fresh process of bin/rtx_example")
    ap.add_argument("--single-process", action="store_true", help="N GPUs from one process through rtx_multi_render")
    ap.add_argument("--time-kernels", action="store_true", help="round-loop path with per-kernel HIP event timing (diagnostic)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # one bare frame under rocprofv3
    ap.add_argument("--force-spawn", action="store_true", help="start the ranks as child processes even for --gpus 1 (exercises the launcher)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend of the ranks; gloo (through host memory) only to rehearse the N-rank path where RCCL cannot run: not a measurement")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: rank r uses GPU r modulo the GPUs there are, so N ranks can run on fewer than N GPUs (needs --backend gloo: RCCL refuses two ranks on one GPU)")
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
    if args.share_gpu:
        if args.backend != "gloo":
            raise SystemExit("--share-gpu needs --backend gloo")
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    through_host = args.backend == "gloo"          # the rehearsal backend moves the buffers through host memory
    cdev = torch.device("cpu") if through_host else dev   # where the bookkeeping tensors of the collectives live
    ranks_seen = 1
    if "WORLD_SIZE" in os.environ and not args.single_process:   # also a world of one, so that the launcher path is the same at every N
        if through_host:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ranks_seen = collective_ranks(cdev)   # also sets the communicator up before anything is timed

    # host side: scene, BVH (SAH PerSurface, the reference's default), camera
    surfaces, cam = make_scene(args.config, args.width, args.spp)
    t0 = time.perf_counter()
    bvh = rt.HostBVH(surfaces, rt.SAH_PER_SURFACE, device=cfg["build"] == "device")
    bvh_ms = (time.perf_counter() - t0) * 1e3
    icam = rt.camera_initialize(cam)
    W, H = icam.image_width, icam.image_height
    nval = W * H * 3
    stream = torch.cuda.current_stream().cuda_stream

    per_rank = None   # N > 1: what every rank did, for the line
    reduce_events = []
    if args.single_process:
        multi = rt.MultiScene(bvh, args.gpus)   # scene on every device + ncclCommInitAll: outside the timed region
        ranks_seen = None
        multi_stats = []

        def step(flags=0):
            nonlocal ranks_seen
            _, ms = multi.render(icam, SEED, flags, to_host=False)
            ranks_seen = ms.collective_ranks
            multi_stats.append(ms)
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
            if world > 1:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if through_host:
                    host = acc.cpu()
                    reduce_accum(host, world)
                    if rank == 0:
                        acc.copy_(host)
                else:
                    reduce_accum(acc, world)
                e1.record()
                reduce_events.append((e0, e1))
            if rank == 0:
                rt.scale(acc.data_ptr(), nval, icam.pixel_samples_scale, image.data_ptr(), stream)
            return st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:   # the reduce's channels for a buffer of this size, before anything is timed (also with --warmup 0)
        reduce_accum(acc.cpu() if through_host else acc, world)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    reduce_events.clear()
    if args.single_process:
        multi_stats.clear()
    barrier()
    t0 = time.perf_counter()
    stats = [step(rt.FLAG_TIME_KERNELS if args.time_kernels else 0) for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    work = torch.tensor([sum(s.rays for s in stats), sum(s.paths for s in stats)], dtype=torch.float64, device=cdev)
    if world > 1:
        # what each rank did: its share of the samples, the device time of its render call and of the reduce as it saw it
        mine = torch.tensor([float(count), sum(s.ms_total for s in stats) / args.steps,
                             sum(a.elapsed_time(b) for a, b in reduce_events) / max(len(reduce_events), 1), elapsed * 1e3 / args.steps],
                            dtype=torch.float64, device=cdev)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank = {"spp_count": [int(v[0].item()) for v in everyone], "ms_render": [round(v[1].item(), 3) for v in everyone],
                    "ms_reduce": [round(v[2].item(), 3) for v in everyone], "ms_step_wall": [round(v[3].item(), 3) for v in everyone]}
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
    elif args.single_process and multi_stats:
        n = args.gpus
        per_rank = {"spp_count": [int(multi_stats[-1].spp_count[g]) for g in range(n)],
                    "ms_render": [round(sum(m.ms_device[g] for m in multi_stats) / len(multi_stats), 3) for g in range(n)],
                    "ms_render_host_wall": [round(sum(m.ms_render[g] for m in multi_stats) / len(multi_stats), 3) for g in range(n)],
                    "ms_reduce": round(sum(m.ms_reduce for m in multi_stats) / len(multi_stats), 3),
                    "ms_frame": round(sum(m.ms_frame for m in multi_stats) / len(multi_stats), 3)}
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
                                                                                     else "one process per GPU (torch.distributed, nccl backend)")
                                if not through_host else
                                f"REHEARSAL, not a measurement: {n_gpus} ranks on {torch.cuda.device_count()} GPU(s), reduce through host memory (gloo)"),
                "ranks_in_collective": ranks_seen,
                "rays_per_frame": int(rays_per_step),
                "paths_per_frame": int(paths_per_step),
                "bvh_build_ms": round(bvh_ms, 2),
            },
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if scene is not None and stats:
            out["config"]["kernel"] = rt.FORM_KERNEL.get(stats[-1].kernel_form)
            # the finished frame on rank 0, as one number: the same at every N up to the order of the f64 additions
            torch.cuda.synchronize()
            out["config"]["image_checksum"] = float(image.sum().item())
        if args.pmc_child or scene is None:   # a bare frame for the counter passes / the single-process form: no counting pass
            print(json.dumps(out), file=json_out, flush=True)
        else:
            if not args.no_image_check and world == 1:
                out["image_check"] = image_check(rt, torch, scene, icam, stats[-1], begin, count, acc, stream)
            out["roofline"] = roofline(rt, scene, bvh.n_nodes, icam, stats, begin, count, acc, stream, args)
            if args.time_kernels:   # round-loop path: summed launch durations per kernel
                out["roofline"]["kernel_time_share"] = {"traverse_ms": round(sum(s.ms_traverse for s in stats) / args.steps, 2),
                                                        "shade_ms": round(sum(s.ms_shade for s in stats) / args.steps, 2),
                                                        "generate_ms": round(sum(s.ms_generate for s in stats) / args.steps, 2)}
            if not args.no_cpu_baseline and world == 1:
                out["cpu_baseline"] = cpu_baseline(args.config, surfaces, cam, cfg["spp"])
                out["speedup_vs_cpu_baseline"] = round(mrays / out["cpu_baseline"]["value"], 1)
            if not args.no_cold_frame and world == 1:
                out["cold_frame"] = cold_frame(args)
            print(json.dumps(out), file=json_out, flush=True)

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def image_check(rt, torch, scene, icam, last, begin, count, acc, stream):
    """The frame of the last timed step (its radiance sums are still in `acc`) against one more render of the same samples by
    ANOTHER implementation of the sample loop: the other form of the single launch where the scene allows both (paths resident
    in registers <-> paths in a pool), else the round loop of separate generate / traverse / shade launches.  Every (pixel,
    sample) has its own random stream, so the two must agree up to the order of the f64 additions and trace the same rays."""
    saved = {k: os.environ.get(k) for k in ("RTX_RESIDENT", "RTX_PERSISTENT")}
    other = torch.zeros_like(acc)
    try:
        os.environ["RTX_RESIDENT"] = "0" if last.kernel_form == rt.FORM_RESIDENT else "1"
        st = scene.render_accumulate(icam, SEED, begin, count, other.data_ptr(), stream)
        if st.kernel_form == last.kernel_form:   # the scene has one form of the single launch only
            os.environ["RTX_PERSISTENT"] = "0"
            other.zero_()
            st = scene.render_accumulate(icam, SEED, begin, count, other.data_ptr(), stream)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    scale = torch.maximum(acc.abs(), other.abs())
    rel = torch.where(scale > 0, (acc - other).abs() / scale.clamp_min(1e-300), torch.zeros_like(acc))
    return {"timed_form": rt.FORM_KERNEL.get(last.kernel_form), "other_form": rt.FORM_KERNEL.get(st.kernel_form),
            "max_rel_diff": float(rel.max().item()), "rays_equal": bool(st.rays == last.rays), "rays": int(st.rays),
            "paths_equal": bool(st.paths == last.paths), "pixels_that_differ_by_more_than_1e-12": int((rel > 1e-12).any(dim=-1).sum().item()),
            "other_form_ms": round(st.ms_total, 3)}


def roofline(rt, scene, n_nodes, icam, stats, begin, count, acc, stream, args):
    """Roofline of the one kernel a frame is (one launch per step: rtx_stats.kernel_form says which), from this rank's launches.

    Work per ray is algorithmic: Nv box tests, Ns sphere tests, Np planar tests per ray of the walk that runs, from a
    counting pass of the same kernel (RTX_FLAG_COUNT_ACTUAL), priced at
        flops = 37 Nv + 30 Ns + 57 Np + 60
        bytes = 56 Nv + 32 Ns + 128 Np (nodes, spheres, planars) + 40 (material) + 160 (ray state, read + write) + 12 (hit record)
                + 12 per path (accumulation).
    The reference's own walk makes more tests on scenes of quads and triangles (it culls with boxes padded to three times
    the surface's extent, src/aabb.rs:78-87): its counts (RTX_FLAG_COUNT_WORK, equal to the CPU oracle's) are reported beside
    (`reference_walk`), and `achieved` is priced on the tests actually made, so that a walk that skips work is not credited
    with it.  Launch time is the HIP-event time of the launch on its stream (rtx_stats.ms_total).
    Which bound applies depends on where the tree lives.  Scenes whose node records sit in LDS (every example scene)
    are bound by f64 vector arithmetic, latency and divergence: `frac` is the f64-VALU fraction, and only the ray state, hit
    records, material records and accumulation of the byte formula can reach HBM (`hbm.algorithmic`).  The 10^6-surface scenes
    walk their tree from global memory: `frac` is the byte formula with the node record this design reads (16 or 32 B per
    visit instead of the 56 B of an SoA f64 node: 48 B box + two links) over the HBM peak; the counters see more (`hbm.measured_over_algorithmic`): every
    16-byte record read that misses L2 moves a whole line."""
    cnt = scene.render_accumulate(icam, SEED, begin, min(count, 2), acc.data_ptr(), stream, rt.FLAG_COUNT_ACTUAL)
    ref = scene.render_accumulate(icam, SEED, begin, min(count, 2), acc.data_ptr(), stream, rt.FLAG_COUNT_WORK)
    nv, ns, npl = cnt.node_visits / cnt.rays, cnt.sphere_tests / cnt.rays, cnt.planar_tests / cnt.rays
    form = stats[-1].kernel_form
    kernel = {rt.FORM_RESIDENT: KERNEL_RESIDENT, rt.FORM_POOLED: KERNEL}.get(form, "k_traverse_f32")
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
    counters, seen = {}, None
    if not args.no_pmc and args.gpus == 1 and form in (rt.FORM_RESIDENT, rt.FORM_POOLED):
        counters, t_rays, seen, note = live_counters(args, kernel)
        if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:   # both in KiB; FETCH_SIZE tallies a 128-byte request as 64 bytes on gfx950
            rd, wr = 2.0 * counters["FETCH_SIZE"] * 1024.0, counters["WRITE_SIZE"] * 1024.0
            traffic = rd + wr
            src = (f"live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one pass each) of {seen} over `python3 bench.py --pmc-child` on this box; "
                   f"read {rd / 1e9:.1f} GB (FETCH_SIZE x 2) + written {wr / 1e9:.1f} GB")
        else:
            src = note
    cfg = CONFIGS[args.config]
    if traffic is None and args.gpus == 1 and (args.spp, args.width) == (cfg["run_spp"], cfg["width"]):
        # (the committed figure is of the configuration's own frame on one GPU: it says nothing about another frame or a rank's share)
        live_note = src
        traffic, t_rays, src = committed_traffic(args.config, kernel)
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
        "kernel": kernel,
        "kernel_seen_by_profiler": seen,
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
        "reference_walk": {"nodes_per_ray": round(ref.node_visits / ref.rays, 3), "sphere_tests_per_ray": round(ref.sphere_tests / ref.rays, 3),
                           "planar_tests_per_ray": round(ref.planar_tests / ref.rays, 3),
                           "flops_per_ray": round(37.0 * ref.node_visits / ref.rays + 30.0 * ref.sphere_tests / ref.rays + 57.0 * ref.planar_tests / ref.rays + 60.0, 1),
                           "note": "box and surface tests of the reference's BVH::hit on the same rays (RTX_FLAG_COUNT_WORK; equal to the CPU oracle's counters); "
                                   "nodes_per_ray etc. above are the tests this walk makes, and `achieved` is priced on those"},
        "box_tests_left_to_f64_tiers": cnt.box_fallbacks / max(cnt.node_visits, 1),
        "rounds_of_busiest_wave": int(max(s.iterations for s in stats)),
        "hbm": hbm,
        "valu_f64": {"achieved_tflops": round(tf, 3), "peak_tflops": FP64_VECTOR_PEAK_TF, "frac": round(tf / FP64_VECTOR_PEAK_TF, 4)},
        "valu_issue": valu_issue(counters, t_rays) if counters else None,
        "on_chip_note": f"{scene_bytes_per_ray:.0f} B/ray of the byte formula are node and surface records "
                        + ("served from LDS (the tree and the surface tables are copied there once per block): not HBM traffic"
                           if tree_on_chip else "read from global memory (the tree is far larger than LDS and L2)")
                        + ("; in the resident form of the kernel the paths stay in registers from bounce to bounce, so the formula's ray state and hit "
                           "records do not move either: what is measured is the image's atomic adds" if form == rt.FORM_RESIDENT else ""),
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
