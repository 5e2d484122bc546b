#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot loop on MI355X.

A step = one pass of the hot path over one frame of a synthetic workload.  Default: BASELINE config 3,
the configuration the metric is quoted on -- the reference's RTOW "final scene" generator (488 spheres, seed 12345),
1920x1080, 512 spp, 50 bounces.  `--config 2|4|5` selects the other GPU configs (parity-test cases, measured for
their own roofline lines).  Scene and BVH are resident in HBM before the timed region and the frame stays in HBM
(`value` = `value_resident`: the bench contract rules a PCIe-inclusive rate out as `value`); `value_e2e` adds the D2H copy of
the float frame, the definition SURVEY 8(d) gives; `scene_setup_ms` / `first_frame_ms` report what sits before the
timed region (BVH build + upload; first-call allocations).

Multi-GPU (strong scaling: the frame is fixed): the image plane is sharded by interleaved 8-row blocks, every rank
renders its blocks and rank 0 gathers the framebuffer slices with ONE RCCL gather.
  * `python bench.py --gpus N` with WORLD_SIZE unset: this process only LAUNCHES -- it starts N ranks with
    torch.distributed.run before anything touches the GPU, relays rank 0's JSON line and exits non-zero if any
    rank fails.
  * under torch.distributed.run (WORLD_SIZE set): one rank per GPU, `torch.distributed` backend "nccl" (= RCCL).
  * `--single-process`: one process drives all N devices through the C-ABI (rtmi_frame_*: ncclCommInitAll +
    one grouped ncclGather inside librtmi.so); with `--force-dist` and N = 1 the frame still builds a one-rank
    communicator and gathers through it.
  * `--force-dist` with `--gpus 1`: the launcher starts ONE rank under torch.distributed.run, which initialises the
    "nccl" (RCCL) process group and runs the frame's gather collective with world size 1 -- the multi-GPU code path on
    the one GPU a box has (`rccl_ranks: 1` in the line).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      algorithmic flops per launch (SURVEY 8d formula, counters from the CPU oracle's instrumented walk
                of the same BVH) / mean kernel duration from HIP events on the launch stream; `traffic`: HBM-side bytes per
                step from FETCH_SIZE / WRITE_SIZE, measured in this run by two rocprofv3 child runs of this command (N = 1)
  cpu_baseline  the oracle (a port of the reference's CPU path, reference-shaped job system) timed on this host
"""
import os

# before anything can initialise HIP/HSA in this process or its children: the pool's host driver only supports dmabuf IPC
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import argparse  # noqa: E402
import re  # noqa: E402
import json  # noqa: E402
import socket  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

SCENE_SEED, RENDER_SEED = 12345, 2025
BLOCK_ROWS = 8
# non-FMA fp32 VALU issue peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (MI355X_MICROARCH.md; its 157.3 TFLOP/s
# vector peak counts an FMA as 2 flops, which the no-contraction parity bar rules out for the reference arithmetic)
VALU_PEAK_TFLOPS = 78.6
SPEC_VECTOR_PEAK_TFLOPS = 157.3  # the guide's fp32 vector peak (packed FMA, 2 flops per lane and clock)

# BASELINE.json configs[1..4].  lin_*: rows the extra linear-scan (reference algorithm) step renders: block k covers rows
# [lin_first + k*lin_stride*8, +8); cpu_stride: the CPU baseline renders every n-th pixel in x and y at full spp
CONFIGS = {
    "2": dict(scene="rtow", width=1200, spp=100, depth=50, lin=None, cpu_stride=1,
              name="RTOW book-1 final scene, 1200x675, 100 spp, 50 bounces"),
    "3": dict(scene="rtow", width=1920, spp=512, depth=50, lin=None, cpu_stride=4,
              name="RTOW final scene 1920x1080, 512 spp, 50 bounces"),
    "4": dict(scene="grid", width=1920, spp=256, depth=50, lin=(128, 34, 4), cpu_stride=64,
              name="100k random spheres with full BVH, 1920x1080, 256 spp"),
    "5": dict(scene="cornell", width=800, spp=4096, depth=200, lin=(96, 25, 4), cpu_stride=8,
              name="Cornell-box-style enclosed scene, 800x800, 4096 spp, 200 bounces"),
}


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes expose all
    host CPUs in the mask but grant a share of them)."""
    hw = os.cpu_count() or 1
    try:
        hw = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    hw = min(hw, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    hw = min(hw, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return hw


def flops_per_sample(ctr):
    """S * (T_sphere * 23 + T_node * 30 + 70), all per-sample means."""
    n = float(ctr["samples"])
    seg = ctr["segments"] / n
    return seg * 70.0 + ctr["sphere_tests"] / n * 23.0 + ctr["node_tests"] / n * 30.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="3")
    ap.add_argument("--width", type=int, default=0, help="override the config's image width (diagnostics)")
    ap.add_argument("--spp", type=int, default=0, help="override the config's samples per pixel (diagnostics)")
    ap.add_argument("--depth", type=int, default=0, help="override the config's bounce limit (diagnostics)")
    ap.add_argument("--accel", choices=("auto", "bvh", "brute"), default="auto",
                    help="auto = the library's own choice (RTMI_ACCEL_AUTO: BVH walk above 24 spheres, the reference's linear scan below)")
    ap.add_argument("--shard-plan", choices=("mod", "cost"), default="mod",
                    help="N > 1: row block b -> rank b mod N (default), or dealt out by the scene's cost map (rtmi_shard_plan)")
    ap.add_argument("--single-process", action="store_true",
                    help="one process, N devices, through rtmi_frame_* (RCCL inside librtmi.so)")
    ap.add_argument("--force-dist", action="store_true",
                    help="world size 1 through the multi-GPU path: launcher, RCCL process group and the gather collective")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-linear-scan", action="store_true", help="skip the extra linear-scan (reference algorithm) step")
    ap.add_argument("--no-e2e", action="store_true", help="skip the extra D2H-inclusive steps behind value_e2e")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 child runs (FETCH_SIZE / WRITE_SIZE) behind roofline.traffic; the figure of "
                         "profiles/hbm_traffic.json is restated instead")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-stride", type=int, default=0, help="CPU baseline renders every n-th pixel in x and y")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="CPU rehearsal of the launcher and the gather plumbing (no render, no GPU, value = 0)")
    ap.add_argument("--rehearse-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--rehearse-height", type=int, default=101, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `--gpus N` without a rendezvous in the environment.  Nothing here imports torch or touches a device.
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Starts args.gpus ranks of this script (one per GPU) as fresh child processes and relays rank 0's JSON line.
    Worker fan-out of the reference: RayTracer::create, src/main.cc:608-712 (one std::thread per core)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"bench.py: a rank failed (torch.distributed.run exit code {proc.returncode})", file=sys.stderr)
        return proc.returncode
    if line is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 3
    doc = json.loads(line)
    if doc.get("n_gpus") != args.gpus:
        print(f"bench.py: asked for {args.gpus} GPUs, the ranks report {doc.get('n_gpus')}", file=sys.stderr)
        return 4
    print(line, flush=True)
    return 0


def measure_traffic(args):
    """roofline.traffic, measured in this run: HBM-side bytes of the trace kernel per step from the PMC counters, collected as
    MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit one), kernel trace
    only beside them -- over child runs of this very script (first frame + one step, nothing else), started BEFORE this process
    touches the GPU.  Returns (bytes per step, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--accel", args.accel, "--steps", "1", "--warmup", "0",
             "--no-cpu-baseline", "--no-linear-scan", "--no-e2e", "--no-traffic", "--traffic-child"]
    for flag, v in (("--width", args.width), ("--spp", args.spp), ("--depth", args.depth)):
        if v:
            child += [flag, str(v)]
    per_frame = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="rtmi_traffic_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.run([prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + child,
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (exit {p.returncode})"
            total, n = 0.0, 0
            for row in csv.DictReader(open(files[0])):
                # (the timed variants only: the scene's one cost probe launch in the first frame is the counting variant,
                # rtmi_trace_kernel<accel, true, ...>)
                if re.search(r"rtmi_trace_kernel<\d+, false,", row.get("Kernel_Name", "")) and row.get("Counter_Name") == counter:
                    total += float(row["Counter_Value"])
                    n += 1
            if n == 0 or n % 2:
                return None, f"unexpected number of trace dispatches in the {counter} pass: {n}"
            per_frame[counter] = (total / 2.0, n // 2)  # the child renders two frames (first frame + one step)
        except Exception as e:  # a profiler hiccup must not cost the bench line
            return None, f"{counter} pass: {type(e).__name__}: {e}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    (f_kb, bands), (w_kb, _) = per_frame["FETCH_SIZE"], per_frame["WRITE_SIZE"]
    note = (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child runs of this command, "
            f"first frame + one step), 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 correction of MI355X_MICROARCH.md) summed over the {bands} "
            f"trace dispatch(es) of one step: FETCH_SIZE {f_kb:.0f} KB, WRITE_SIZE {w_kb:.0f} KB.  WRITE_SIZE tallies 64 B per write request "
            f"and the sample records leave as lone 16-byte stores: it reads 3.65x their bytes (tools/ubench/write_size_calib.hip)")
    return int((2.0 * f_kb + w_kb) * 1024.0), note


def workload(pkg, cfg, args):
    """(camera params, objects, materials) of a BASELINE config."""
    width, spp, depth = args.width or cfg["width"], args.spp or cfg["spp"], args.depth or cfg["depth"]
    if cfg["scene"] == "rtow":
        objs, mats = pkg.make_world_spheres(SCENE_SEED)
        kw = dict(image_width=width, samples_per_pixel=spp, max_depth=depth)
        label = f"S-RTOW(seed {SCENE_SEED})"
    elif cfg["scene"] == "grid":
        objs, mats, kw = pkg.workloads.big_grid(316)
        kw.update(image_width=width, samples_per_pixel=spp, max_depth=depth)
        label = "S-GRID(316x316 jittered spheres over a ground sphere, seed 4)"
    else:
        objs, mats, kw = pkg.workloads.cornell_like()
        kw.update(image_width=width, samples_per_pixel=spp, max_depth=depth)
        label = "S-CORNELL(5 wall spheres R=1e3, glass + metal sphere)"
    return kw, objs, mats, label


def rehearse(args, world, rank):
    """Launcher + gather plumbing on CPU tensors (gloo): every rank fills its slice with the absolute row number, rank 0
    gathers and checks scanline order.  No render, no GPU; the line says so."""
    import torch
    import torch.distributed as dist
    import rtmi_loader
    pkg = rtmi_loader.load()
    if world > 1:
        dist.init_process_group(os.environ.get("RTMI_DIST_BACKEND", "gloo"))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    if rank == args.rehearse_fail_rank:
        raise SystemExit(7)
    H, W = args.rehearse_height, 16
    plan = pkg.RowShardPlan(H, BLOCK_ROWS, world)
    y_first, n_blocks, rows = plan.shard(rank)
    local = torch.full((plan.max_rows, W, 3), -1.0)
    loc = 0
    for k in range(n_blocks):
        y0 = y_first + k * world * BLOCK_ROWS
        for y in range(y0, min(H, y0 + BLOCK_ROWS)):
            local[loc] = float(y)
            loc += 1
    frame = pkg.gather_frame(local, plan, rank)
    if rank == 0:
        assert torch.equal(frame[:, 0, 0], torch.arange(H, dtype=torch.float32)), "gathered rows out of order"
        print(json.dumps({"metric": "launcher rehearsal (no render)", "value": 0.0, "unit": "Msamples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "none"}, "rehearsal": True, "rows": H,
                          "blocks_per_rank": [s_[1] for s_ in plan.shards],
                          "ranks": world, "backend": dist.get_backend() if world > 1 else "none"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if (args.gpus > 1 or args.force_dist) and world_env is None and not args.single_process:
        sys.exit(launch_ranks(args, argv))  # this process never touches the GPU

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_process:
        if world != 1:
            raise SystemExit("--single-process runs without torch.distributed.run")
    elif world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse_launch:
        return rehearse(args, world, rank)

    live_traffic = (None, "not measured (--no-traffic, a multi-GPU run or --single-process): the figure of profiles/hbm_traffic.json")
    if world == 1 and not args.single_process and not args.no_traffic and not args.force_dist:
        live_traffic = measure_traffic(args)  # child processes; this one has not touched the GPU yet

    import numpy as np
    import torch
    import torch.distributed as dist
    import rtmi_loader
    pkg = rtmi_loader.load()

    backend = os.environ.get("RTMI_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0  # rehearsal: every rank on the one visible GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    n_gpus = args.gpus
    if args.single_process and torch.cuda.device_count() < n_gpus:
        raise SystemExit(f"--single-process --gpus {n_gpus} but {torch.cuda.device_count()} devices are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or (args.force_dist and not args.single_process)
    if use_dist:
        # RCCL ("nccl") over xGMI is the product path; RTMI_DIST_BACKEND=gloo lets two ranks share ONE GPU for a
        # rehearsal of everything but the collective itself (RCCL refuses two ranks on one device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    cfg = CONFIGS[args.config]
    kw, objs, mats, label = workload(pkg, cfg, args)
    cam = pkg.camera_setup(pkg.camera_params(**kw))
    W, H, spp, depth = cam.img_width, cam.img_height, cam.samples_per_pixel, cam.maxdepth
    # RTMI_ACCEL_AUTO's rule, restated so that the label is known before the scene exists: the walk overtakes the scan
    # between 20 and 30 spheres (tools/scan_crossover.py; the 7-sphere box of config 5 is 1.3x faster scanned)
    if args.accel == "auto":
        args.accel = "bvh" if len(objs) > 24 else "brute"
    accel = pkg.ACCEL_BVH if args.accel == "bvh" else pkg.ACCEL_BRUTE
    samples = W * H * spp

    stream = torch.cuda.current_stream(dev).cuda_stream
    kernel_ms = []
    gather_ms = []
    gather_events = []  # torch.distributed path: (before, after) events around the gather + de-interleave of every step
    t_setup = time.perf_counter()
    if args.single_process:
        # (librccl prints a version banner on stdout when the first communicator is created: this process owes its stdout
        # ONE JSON line, so the banner goes to stderr)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            frame_obj = pkg.Frame(cam, objs, mats, devices=tuple(range(n_gpus)), block_rows=BLOCK_ROWS, accel=accel,
                                  force_rccl=args.force_dist)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        scene = None

        def step():
            frame_obj.render_device(RENDER_SEED)
            t = frame_obj.timing()
            kernel_ms.append(t["kernel_ms"])
            gather_ms.append(t["gather_ms"])
            return None, None
    else:
        scene = pkg.Scene(cam, objs, mats, accel=accel, device=local_rank)
        torch.cuda.synchronize(dev)
        scene_setup_ms = (time.perf_counter() - t_setup) * 1e3
        # row blocks -> ranks: block b -> rank b mod N, or (--shard-plan cost) by the scene's cost map, longest processing
        # time first.  Every rank's probe counts the same integers; rank 0's costs travel all the same, so that no rank can disagree.
        # (Measured on the one-GPU rehearsal, profiles/r06_shard_perf.txt: the two plans' slowest shards are 17.88 and 17.86 ms at
        # N = 8 -- segment counts balance to 0.1 % and the shards stay 4 % apart -- so the default stays the plan of rounds 1-5.)
        costs = scene.tile_costs() if (world > 1 and args.shard_plan == "cost") else None
        bc = pkg.block_costs(costs, H, BLOCK_ROWS) if costs is not None else np.zeros((H + BLOCK_ROWS - 1) // BLOCK_ROWS, np.uint64)
        if use_dist and args.shard_plan == "cost":  # (the default plan needs no exchange)
            t_bc = torch.from_numpy(bc.astype(np.int64)).to(dev if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(t_bc, src=0)
            bc = t_bc.cpu().numpy().astype(np.uint64)
        plan = pkg.CostShardPlan(H, BLOCK_ROWS, world, bc if bc.any() else None)
        my_blocks = plan.blocks(rank)
        n_blocks, rows = len(my_blocks), plan.rows(rank)
        # one buffer per rank -- float RGB then RGBA8 (as bits) -- so that the frame travels in ONE collective
        n_px = plan.max_rows * W
        local = torch.zeros(n_px * 4, dtype=torch.float32, device=dev)
        rgb = local[:n_px * 3].view(plan.max_rows, W, 3)
        rgba = local[n_px * 3:].view(torch.int32).view(plan.max_rows, W, 1)

        gathered = torch.empty(world * n_px * 4, dtype=torch.float32, device=dev) if rank == 0 else None
        gathered_parts = list(gathered.view(world, n_px * 4).unbind(0)) if rank == 0 else None

        def split(parts):
            """[world * n_px * 4] gathered floats -> (rgb frame, rgba frame) in scanline order"""
            per = parts.view(world, n_px * 4)
            idx = torch.as_tensor(plan.index, device=parts.device)
            f = per[:, :n_px * 3].reshape(world * plan.max_rows, W, 3).index_select(0, idx)
            f8 = per[:, n_px * 3:].reshape(world * plan.max_rows, W, 1).view(torch.int32).index_select(0, idx)
            return f, f8

        def step():
            if n_blocks and (world == 1 or not bc.any()):  # block b -> rank b mod N: the strided set of rounds 1-5
                scene.render_row_blocks_device(rank * BLOCK_ROWS, BLOCK_ROWS, world, n_blocks, RENDER_SEED, rgb.data_ptr(), rgba.data_ptr(), stream)
            elif n_blocks:
                scene.render_block_list_device(BLOCK_ROWS, my_blocks, RENDER_SEED, rgb.data_ptr(), rgba.data_ptr(), stream)
            if not use_dist:
                return rgb[:H], rgba[:H]
            if dist.get_backend() != "nccl":  # rehearsal backend: stage through host memory
                torch.cuda.synchronize(dev)
                host = local.cpu()
                got = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, got, dst=0)
                return split(torch.cat(got).to(dev)) if rank == 0 else (None, None)
            # the one RCCL collective of a frame, straight into one preallocated buffer on rank 0 (rank-major slices);
            # bracketed by events on the stream it runs on, so that the line can say what the gather cost
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            dist.gather(local, gathered_parts, dst=0)
            out = split(gathered) if rank == 0 else (None, None)
            g1.record()
            gather_events.append((g0, g1))
            return out

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.single_process:
        scene_setup_ms = (time.perf_counter() - t_setup) * 1e3
    # the first frame of a scene also allocates its sample-record buffer (16 B per sample of the call): timed on its own,
    # outside the warm-up count
    t_first = time.perf_counter()
    step()
    torch.cuda.synchronize(dev)
    my_first_frame_ms = (time.perf_counter() - t_first) * 1e3  # this rank's own first frame, before the barrier
    sync()
    first_frame_ms = (time.perf_counter() - t_first) * 1e3
    for _ in range(args.warmup):
        step()
    sync()
    kernel_ms.clear()
    gather_ms.clear()
    gather_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame, frame8 = step()
        # per-launch duration from the HIP events the library records on the launch stream (blocks on that launch;
        # the next step cannot start earlier anyway because it reuses the same output buffers)
        if scene is not None and n_blocks:
            kernel_ms.append(scene.last_kernel_ms())
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if gather_events:  # (all steps are done: the events have completed)
        gather_ms.extend(a.elapsed_time(b) for a, b in gather_events)
        gather_events.clear()
    # what every rank saw, for a line that explains itself when N > 1: its own first frame (allocations, the cost
    # probe), its mean trace-kernel span, its mean gather + de-interleave time (rank 0: until the frame is in scanline order;
    # the others: until their slice has left)
    first_frame_per_rank = [my_first_frame_ms]
    gather_per_rank = [float(np.mean(gather_ms))] if gather_ms else None
    if use_dist:
        mine = torch.tensor([my_first_frame_ms, float(np.mean(gather_ms)) if gather_ms else 0.0], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        first_frame_per_rank = [float(v[0].item()) for v in allv]
        gather_per_rank = [float(v[1].item()) for v in allv]
    # mean trace-kernel time of every rank / device
    if args.single_process:
        per_rank_ms = [float(np.mean([k[i] for k in kernel_ms])) for i in range(n_gpus)] if kernel_ms else [0.0] * n_gpus
    else:
        kmean = torch.tensor([sum(kernel_ms) / max(1, len(kernel_ms))], dtype=torch.float64, device=dev)
        if use_dist:
            allk = [torch.zeros_like(kmean) for _ in range(world)]
            dist.all_gather(allk, kmean)
            per_rank_ms = [float(k.item()) for k in allk]
        else:
            per_rank_ms = [float(kmean.item())]

    if args.traffic_child:  # a profiled child of measure_traffic(): the frames are rendered, nothing else is wanted
        print(json.dumps({"traffic_child": True, "steps": args.steps, "kernel_ms": per_rank_ms}), flush=True)
        scene.close()
        return
    # ---- the metric as SURVEY 8(d) words it: wall time including the D2H copy of the float frame (+ gather) ------------
    e2e_elapsed, e2e_steps = None, 0
    if not args.no_e2e and not args.single_process:
        e2e_steps = max(1, min(args.steps, 3))
        host = torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True) if rank == 0 else None
        sync()
        t1 = time.perf_counter()
        for _ in range(e2e_steps):
            f, _f8 = step()
            if rank == 0:
                host.copy_(f[:H], non_blocking=True)
            torch.cuda.synchronize(dev)
        sync()
        te = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
        e2e_elapsed = float(te.item())
    elif args.single_process and not args.no_e2e:
        e2e_steps = max(1, min(args.steps, 3))
        t1 = time.perf_counter()
        for _ in range(e2e_steps):
            host_rgb, _ = frame_obj.render(RENDER_SEED)  # rtmi_frame_render: gather + D2H into host buffers
        e2e_elapsed = time.perf_counter() - t1

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = samples / (elapsed / args.steps) / 1e6
        if args.single_process:
            launcher = "one process, rtmi_frame_* (ncclCommInitAll + one grouped ncclGather inside librtmi.so)"
            rccl_ranks = frame_obj.rccl_ranks
        else:
            launcher = ("torch.distributed.run, one rank per GPU, backend %s" % dist.get_backend()) if use_dist else "one process"
            rccl_ranks = world if (use_dist and dist.get_backend() == "nccl") else 0
        out = {
            "metric": "Msamples/sec (W*H*spp/s), RTOW final scene" if cfg["scene"] == "rtow" else "Msamples/sec (W*H*spp/s)",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label} {len(objs)} spheres, {W}x{H}, {spp} spp, {depth} bounces, accel={args.accel}",
                       "baseline_config": f"configs[{int(args.config) - 1}]: {cfg['name']}",
                       "sharding": (f"{BLOCK_ROWS}-row blocks dealt out by the scene's cost map (longest processing time first, rtmi_shard_plan) x{n_gpus}, 1 gather to rank 0"
                                    if (not args.single_process and world > 1 and bc.any()) else f"interleaved {BLOCK_ROWS}-row blocks x{n_gpus}, 1 gather to rank 0"),
                       "launcher": launcher,
                       "resident": "scene+BVH in HBM before the timed region; frame stays in HBM"},
            "rccl_ranks": rccl_ranks,
            "kernel_ms_per_rank": [round(k, 3) for k in per_rank_ms],
            "kernel_ms_note": "per rank: trace kernels of a step (HIP events on the launch stream; bands added up, resolve passes not "
                              "included), mean over the timed steps",
            "kernel_ms_slowest_fastest": [round(max(per_rank_ms), 3), round(min(per_rank_ms), 3)],
            "first_frame_ms_per_rank": [round(v, 2) for v in first_frame_per_rank],
            # what `value` is: the bench contract wants inputs resident and rules a PCIe-inclusive rate out as `value`;
            # SURVEY 8(d)'s wording of the metric (D2H of the frame inside the clock) is `value_e2e` below
            "value_resident": round(value, 2),
            "value_definition": "W*H*spp / wall time of the K timed steps, scene + BVH resident, frame left in HBM (gather included "
                                "for N > 1); value_e2e = the same with the D2H copy of the float frame inside the clock",
            "scene_setup_ms": round(scene_setup_ms, 2),
            "scene_setup_note": "rtmi_scene_create on rank 0: SAH BVH build on the host + upload of spheres, materials and nodes",
            "first_frame_ms": round(first_frame_ms, 2),
            "first_frame_note": "first call on the scene (allocates the sample-record buffer, 16 B per sample of the call); not "
                                "one of the W warm-up or K timed steps",
        }
        if gather_ms:
            out["gather_ms"] = round(float(np.mean(gather_ms)), 3)
        if gather_per_rank is not None and use_dist:
            out["gather_ms_per_rank"] = [round(v, 3) for v in gather_per_rank]
            out["gather_ms_note"] = ("torch.distributed path: events around dist.gather + de-interleave on the rank's stream, behind its "
                                     "own kernels; rank 0's figure includes waiting for the slowest rank's slice")
        if e2e_elapsed is not None:
            out["value_e2e"] = round(samples / (e2e_elapsed / e2e_steps) / 1e6, 2)
            out["value_e2e_note"] = (f"{e2e_steps} extra steps timed with the D2H copy of the {W}x{H} float frame into pinned "
                                     "host memory (and the gather) inside the clock: the metric as SURVEY 8(d) words it")
        if args.single_process:
            frame = torch.from_numpy(host_rgb) if e2e_elapsed is not None else None
        # ---- parity spot check + algorithmic work counters from the CPU oracle (checker only) -------------
        from oracle import binding as ob
        ocam = ob.camera_setup(ob.camera_params(**kw))
        rng = np.random.default_rng(1)
        n_chk = 48 if cfg["scene"] == "rtow" else 12
        if frame is not None:
            frame_h = frame.cpu().numpy()
            xs, ys = rng.integers(0, W, n_chk), rng.integers(0, H, n_chk)
            obvh = None
            if len(objs) > 4096:  # the oracle's own linear scan over 100k spheres takes minutes per pixel: walk the BVH
                obvh = pkg.bvh_build(objs)
                obvh = dict(obvh, nodes=obvh["nodes"].view(ob.BVH_NODE_DTYPE))
            worst = 0.0
            for x, y in zip(xs, ys):
                want, _ = ob.render_rect_counter(ocam, objs, mats, RENDER_SEED, int(x), int(y), int(x) + 1, int(y) + 1,
                                                 nthreads=1, bvh=obvh)
                d = np.abs(frame_h[y, x] - want[0, 0])
                d[np.isnan(frame_h[y, x]) & np.isnan(want[0, 0])] = 0.0  # the reference arithmetic can yield a NaN pixel
                worst = max(worst, float(d.max()))
            out["parity_check"] = {"pixels": n_chk, "max_abs_diff_vs_oracle": worst,
                                   "oracle": "linear scan" if obvh is None else "instrumented BVH walk (== linear scan, tests)"}
        # the tree the kernel walks (the library's default leaf size depends on where the scene lives)
        bvh = None
        if args.accel == "bvh":
            bvh = scene.bvh() if scene is not None else pkg.bvh_build(objs, 4 if len(objs) > 0x2000 else 2)
            bvh = dict(bvh, nodes=bvh["nodes"].view(ob.BVH_NODE_DTYPE))
        ctr_stride = 24
        ctr = {"samples": 0, "segments": 0, "sphere_tests": 0, "node_tests": 0, "hit_lambertian": 0, "hit_metallic": 0}
        sub_spp = min(spp, 64)
        ccam = ob.camera_setup(ob.camera_params(**dict(kw, samples_per_pixel=sub_spp)))
        for y in range(ctr_stride // 2, H, ctr_stride):
            for x in range(ctr_stride // 2, W, ctr_stride * 4):
                _, _, c = ob.render_rect_counter(ccam, objs, mats, RENDER_SEED, x, y, x + 1, y + 1, counters=True,
                                                 bvh=bvh)
                for k in ctr:
                    ctr[k] += c[k]
        fps = flops_per_sample(ctr)
        # the same count for walks that all start at the root (rounds 1-5's algorithm): camera entries and walk starts make the
        # ALGORITHM cheaper, so `roofline.frac` -- priced on the work of the walk as it runs, as SURVEY 8(d) asks -- can fall while the
        # kernel gets faster; frac_at_root_walk_work keeps the yardstick of the earlier rounds' lines beside it
        fps_root = fps
        if bvh is not None and (bvh.get("entries") is not None or bvh.get("walk_starts") is not None):
            plain = dict(bvh, entries=None, walk_starts=None)
            ctr0 = {k: 0 for k in ctr}
            for y in range(ctr_stride // 2, H, ctr_stride):
                for x in range(ctr_stride // 2, W, ctr_stride * 4):
                    _, _, c = ob.render_rect_counter(ccam, objs, mats, RENDER_SEED, x, y, x + 1, y + 1, counters=True, bvh=plain)
                    for k in ctr0:
                        ctr0[k] += c[k]
            fps_root = flops_per_sample(ctr0)
        # (what was launched, not what the scene is eligible for: a launch whose chain slots could not be allocated falls back
        # to run-length encoded chains and says so in packed_chain_fallbacks)
        li = scene.launch_info() if scene is not None else {}
        if li:
            out["launch"] = {"bands": li.get("bands"), "tile_order": li.get("tile_order"), "probe_us": li.get("probe_us"),
                             "cam_entry": li.get("cam_entry"), "entry_build_us": li.get("entry_build_us"),
                             "note": "the most recent call: bands of rows it was rendered in, 8x8 tiles handed out costliest first (1) or row "
                                     "by row (0), duration of the scene's one cost probe launch and host time of its table of camera-ray "
                                     "entries (both inside scene_setup_ms since round 6)"}
        chain_words = li.get("packed_chains", 0) if li.get("packed_chain_fallbacks", 0) == 0 else 0
        chain_bytes = 0.0
        if chain_words:
            bits = max(1, int(np.ceil(np.log2(max(2, len(mats))))))
            chain_bytes = (ctr["hit_lambertian"] + ctr["hit_metallic"]) / ctr["samples"] * bits / 8.0
        kernel_s = max(per_rank_ms) / 1e3
        samples_per_launch = samples / n_gpus
        achieved = samples_per_launch * fps / kernel_s / 1e12
        traffic, traffic_note = live_traffic
        traffic_source = "rocprofv3 child runs of this command, in this run" if traffic is not None else None
        tpath = os.path.join(_ROOT, "profiles", "hbm_traffic.json")
        if traffic is None and os.path.exists(tpath):
            try:
                for tj in json.load(open(tpath)).get("entries", []):
                    if (tj.get("config") == args.config and tj.get("width") == W and tj.get("spp") == spp
                            and tj.get("n_gpus") == n_gpus and tj.get("accel", "bvh") == args.accel):
                        traffic = tj.get("bytes_per_launch")
                        traffic_note = (tj.get("note") or "") + f"  [{live_traffic[1]}]"
                        traffic_source = "profiles/hbm_traffic.json (the builder's rocprofv3 passes of this command, restated)"
            except Exception:
                traffic = None
        # the same work in lane-operations (what the VALU issues): a box test is 6 FMA + 18 single operations = 24, a
        # sphere test 23, a segment's shading 70; and against the guide's packed-FMA vector peak
        n_s = float(ctr["samples"])
        lane_ops = ctr["segments"] / n_s * 70.0 + ctr["sphere_tests"] / n_s * 23.0 + ctr["node_tests"] / n_s * 24.0
        out["roofline"] = {
            "bound": "valu", "achieved": round(achieved, 4), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / VALU_PEAK_TFLOPS, 5),
            "frac_vs_spec_vector_peak": round(achieved / SPEC_VECTOR_PEAK_TFLOPS, 5),
            "spec_vector_peak": SPEC_VECTOR_PEAK_TFLOPS,
            "lane_op_frac": round(samples_per_launch * lane_ops / kernel_s / 1e12 / VALU_PEAK_TFLOPS, 5),
            "lane_ops_per_sample": round(lane_ops, 1),
            "traffic": traffic, "traffic_source": traffic_source, "traffic_note": traffic_note,
            "kernel": "rtmi_trace_kernel<%s>" % args.accel, "kernel_ms": round(kernel_s * 1e3, 3),
            "flops_per_sample": round(fps, 1),
            "flops_per_sample_root_walk": round(fps_root, 1),
            "frac_at_root_walk_work": round(samples_per_launch * fps_root / kernel_s / 1e12 / VALU_PEAK_TFLOPS, 5),
            "frac_note": "frac = the work of the walk AS IT RUNS (camera rays from their tile's entry, scattered rays of HBM-resident trees "
                         "from their own leaf: fewer box tests per sample than a walk from the root); frac_at_root_walk_work = the same "
                         "time against the work of rounds 1-5's walk from the root, the yardstick of the earlier rounds' lines",
            "counters_per_sample": {k: round(ctr[k] / ctr["samples"], 3) for k in ctr if k not in ("samples", "hit_lambertian", "hit_metallic")},
            "counters_source": f"oracle instrumented walk, {ctr['samples']} samples on a uniform pixel subset",
            # scene staged once per workgroup-resident CU + 16-byte sample records written once (the ordered resolve
            # pass reads them back) + framebuffer slice; the path is VALU-bound, HBM is reported as a sanity check
            # + in packed-chain launches the material handles a path leaves for the resolve pass (its non-dielectric bounces x
            # ceil(log2 n_materials) bits; the slot the launch reserves per sample is chain_words x 4 bytes)
            "hbm_algorithmic_bytes_per_launch": int(len(objs) * (16 + 16 + 32) + (0 if bvh is None else len(bvh["nodes"]) * 48)
                                                    + samples_per_launch * (16 + chain_bytes) + W * (H // n_gpus) * 16),
            "chain_words_per_sample": chain_words,
            "note": "fp32 VALU-bound path (SURVEY 8d): peak = non-FMA issue rate 256 CU x 4 SIMD x 32 lanes x 2.4 GHz",
        }
        # ---- the reference's own algorithm on the GPU: linear closest-hit scan on the same frame (or a uniform subset of
        # its row blocks where the whole frame would take minutes), outside the timed region ----------------------------
        lin_rate = None
        if world == 1 and not args.single_process and args.accel == "bvh" and not args.no_linear_scan:
            y_lin, s_lin, n_lin = cfg["lin"] or (0, 1, (H + BLOCK_ROWS - 1) // BLOCK_ROWS)
            if args.width or H <= y_lin + (n_lin - 1) * s_lin * BLOCK_ROWS:
                y_lin, s_lin, n_lin = 0, 1, (H + BLOCK_ROWS - 1) // BLOCK_ROWS
            lin_rows = sum(min(BLOCK_ROWS, H - (y_lin + k * s_lin * BLOCK_ROWS)) for k in range(n_lin))
            want = torch.zeros((lin_rows, W, 3), dtype=torch.float32, device=dev)
            scene.render_row_blocks_device(y_lin, BLOCK_ROWS, s_lin, n_lin, RENDER_SEED, want.data_ptr(), 0, stream)
            torch.cuda.synchronize(dev)
            got = torch.zeros_like(want)
            lin = pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BRUTE, device=local_rank)
            lin_ms = None
            for _ in range(2 if cfg["lin"] is None else 1):
                lin.render_row_blocks_device(y_lin, BLOCK_ROWS, s_lin, n_lin, RENDER_SEED, got.data_ptr(), 0, stream)
                torch.cuda.synchronize(dev)
                lin_ms = lin.last_kernel_ms()
            lin.close()
            same = bool(torch.equal(torch.nan_to_num(got).view(torch.int32), torch.nan_to_num(want).view(torch.int32)))
            lin_samples = lin_rows * W * spp
            lin_fps = ctr["segments"] / ctr["samples"] * (len(objs) * 23.0 + 70.0)
            lin_rate = lin_samples / lin_ms / 1e3
            out["linear_scan_kernel"] = {
                "value": round(lin_rate, 3), "unit": "Msamples/s", "kernel_ms": round(lin_ms, 3),
                "rows": f"{lin_rows} of {H} (blocks of {BLOCK_ROWS} rows, every {s_lin}th from row {y_lin})",
                "flops_per_sample": round(lin_fps, 1),
                "roofline_frac": round(lin_samples * lin_fps / (lin_ms / 1e3) / 1e12 / VALU_PEAK_TFLOPS, 4),
                "frame_bit_identical_to_bvh": same,
                "note": "rtmi_trace_kernel<brute>: the reference's O(N) scan (object.defs.cc:68-81); the BVH walk returns "
                        "the same frame with far less algorithmic work",
            }
            # the normaliser for lines from different boxes: the walk's kernel time per sample over the scan's, both
            # measured in this run on this box (the scan is 94 % of its roofline and its code moves little between rounds)
            out["walk_over_scan_kernel_ratio"] = round((max(per_rank_ms) / samples) / (lin_ms / lin_samples), 5)
        # ---- CPU baseline: the oracle, reference-shaped job system, on this host's cores -----------------------
        if world == 1 and not args.single_process and not args.no_cpu_baseline:
            hw = usable_cpus()
            threads = hw - 2 if hw > 6 else hw  # src/main.cc:608-611
            stride = args.cpu_stride or cfg["cpu_stride"]
            secs, n = ob.bench_mt(ocam, objs, mats, SCENE_SEED, stride, threads)
            out["cpu_baseline"] = {
                "value": round(n / secs / 1e6, 5), "unit": "Msamples/s", "cores": threads, "kind": "port",
                "sample": f"every {stride}th pixel in x and y of the same frame at full spp "
                          f"({n} samples, {secs:.1f} s); the reference's linear scan, mt19937 per worker, shuffled 8x8 "
                          f"tiles (src/main.cc:608-633)",
                "host_cpus": hw,
            }
            if cfg["scene"] == "rtow":
                secs1, n1 = ob.bench_mt(ocam, objs, mats, SCENE_SEED, stride * 6, 1)
                out["cpu_baseline"]["single_thread_value"] = round(n1 / secs1 / 1e6, 5)
            out["gpu_over_cpu"] = round(value / (n / secs / 1e6), 1)
            out["gpu_over_cpu_note"] = "GPU: BVH walk (same frame as the scan, bit for bit); CPU: the reference's linear scan"
            if lin_rate is not None:
                out["gpu_over_cpu_like_for_like"] = round(lin_rate / (n / secs / 1e6), 1)
        print(json.dumps(out), flush=True)

    if scene is not None:
        scene.close()
    if args.single_process:
        frame_obj.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
