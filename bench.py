#!/usr/bin/env python3
"""Throughput benchmark of the path-trace hot path.

    python bench.py --gpus N --steps K --warmup W

One *step* = one render pass of `--spp-per-step` samples per pixel per GPU over
the whole frame (default workload: the reference Cornell box at 1920x1080,
512 spp per GPU per step — with 8 GPUs one step is the 4096-spp frame of the
headline configuration, with one GPU it is one GPU's share of it), followed for
N > 1 by the RCCL sum-reduce of the fp32 accumulators to rank 0 and, on rank 0,
the resolve to 8 bits; for N > 1 reduce and resolve of a step run on a second stream while the
next step renders (two accumulators alternate), a single GPU runs them in line.  Ranks render disjoint absolute sample ranges (weak
scaling: per-GPU work is fixed); the random stream is keyed by the absolute
sample index, so the image does not depend on N beyond fp32 summation order.

The scene is resident in HBM before the timed region; the accumulator never
leaves the device inside it (only the 8-bit resolve is copied to the host).

`--gpus N` with N > 1 and no launcher (no WORLD_SIZE in the environment) starts the N ranks
itself, as child processes, and forwards rank 0's line; under torchrun (`python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N`) it is one of the ranks.
`--single-process` drives all N GPUs from one process through rtx_render_multi instead.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X FP32 vector peak: 256 CU x 4 SIMD x 32 FMA lanes x 2 flops x 2.4 GHz = 157.3 TFLOP/s.
# Measured on the part (tools/probe/valu_cost.hip): a wave64 v_fma_f32 with VGPR operands issues
# every ~2.5 cycles and a v_pk_fma_f32 every ~4.5, so plain and packed FMAs reach the same rate.
PEAK_FP32_VALU_TFLOPS = 157.3

WORKLOADS = {
    # name: (scene, width, height, default spp per step)
    "cornell_1080p": ("cornell", 1920, 1080, 512),
    # the two BVH configurations: one step = the 1024-spp frame on every GPU.  A wave owns 4x4 pixels x spp
    # items, and its lanes idle while the last paths of that pool finish: launches of 128 / 256 spp run 11 %
    # (100 000-item scene) and 4 % (sphere field) slower than the whole frame in one launch.
    "random_field_1080p": ("random_field", 1920, 1080, 1024),
    "bvh_stress_1080p": ("bvh_stress", 1920, 1080, 1024),
    "cornell_extra_4k": ("cornell_extra", 3840, 2160, 256),
    "small_400x225": ("small", 400, 225, 100),
}


# Samples per pixel of one frame (or progressive step) of each configuration as a whole, which N GPUs split
# N ways (--share-of N): the 4096-spp Cornell frame, the 1024-spp BVH frames, the 256-spp steps of the 4K frame.
FRAME_SPP = {"cornell_1080p": 4096, "random_field_1080p": 1024, "bvh_stress_1080p": 1024, "cornell_extra_4k": 256,
             "small_400x225": 100}


def load_flops_per_ray(scene: str, bvh: bool):
    """Algorithmic fp32 operations per ray, counted by the op-counting build of the CPU
    restatement (tests/golden/flops_per_ray.json, made by tests/golden/make_golden.py): the
    reference's O(N) scan for scenes rendered by the scan; for scenes rendered through the tree,
    the same rays through the renderer's own tree walked in the kernel's order (key ":bvh_product":
    24 per inner node visited + the item tests reached + shading).  Returns (flops per ray, source,
    flops per ray through the restatement's own median-split tree or None)."""
    path = os.path.join(ROOT, "tests", "golden", "flops_per_ray.json")
    try:
        with open(path) as f:
            table = json.load(f)
        if not bvh:
            return float(table[scene]["flops_per_ray"]), "counted (the reference's linear scan)", None
        other = table.get(scene + ":bvh")
        return (float(table[scene + ":bvh_product"]["flops_per_ray"]),
                "counted (the renderer's own tree walked in the kernel's order by the CPU restatement)",
                float(other["flops_per_ray"]) if other else None)
    except Exception:
        return None, f"no count for {scene} ({'tree' if bvh else 'scan'})", None


def load_hbm_traffic(workload: str, spp: int):
    """HBM bytes per launch of the trace kernel from the PMC passes recorded under profiles/
    (tools/profile_bench.sh); None when no measurement exists for this exact workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            rec = json.load(f)[workload]
        return int(rec["bytes_per_launch"]) if rec.get("spp_per_launch") == spp else None
    except Exception:
        return None


def cpu_baseline(scene: str, w: int, h: int, threads: int, seconds_budget: float = 20.0) -> dict:
    """The reference tracer itself (oracle/_ref) timed on this host's cores on a bounded sample
    of the same workload; beside it the CPU restatement (no shared_ptr traffic in the hit
    record, so it scales with the cores where the reference does not).  Without oracle/_ref
    the restatement is the baseline (kind "port").

    The sample is the workload's scene and camera at the largest frame and sample count that
    fit the time budget: the full frame at 1 .. 64 spp where one sample per pixel is affordable
    (the Cornell box), a frame scaled down in both axes at 1 spp where it is not (the reference
    has no acceleration structure: a ray of the 100 000-item scene costs it a millisecond)."""
    from oracle import ref, restate
    from raytrace_amd import api

    def scene_at(fw, fh):
        flat = api.HostScene(scene, fw, fh)
        rs = ref.RefScene(scene, fw, fh) if ref.available() else None
        return flat, rs

    def run_port(flat, fw, fh, spp, n_threads):
        t = time.perf_counter()
        out = restate.render(flat, fw, fh, 0, spp, seed=0, threads=n_threads)
        return time.perf_counter() - t, out["rays"]

    kind = "reference" if ref.available() else "port"

    def run(flat, rs, fw, fh, spp, n_threads):
        if rs is None:
            return run_port(flat, fw, fh, spp, n_threads)
        t = time.perf_counter()
        out = rs.render(0, spp, seed=0, threads=n_threads)
        return time.perf_counter() - t, out["rays"]

    def size_for(samples):
        """The frame (aspect of the workload's) and sample count for about `samples` samples."""
        if samples >= w * h:
            return w, h, int(max(1, min(64, samples / (w * h))))
        f = max((samples / (w * h)) ** 0.5, 16.0 / w)
        return max(16, int(w * f)), max(9, int(h * f)), 1

    scenes = {}

    def scene_cached(fw, fh):
        if (fw, fh) not in scenes:
            scenes[(fw, fh)] = scene_at(fw, fh)
        return scenes[(fw, fh)]

    def calibrated(budget, n_threads, port=False):
        """Run at growing sizes - a 16x9 probe, about two seconds, the budget - each sized from the
        rate of the one before (tiny frames under-report the rate); returns the last run."""
        fw, fh, spp = 16, 9, 1
        for target in (None, 2.0, budget):
            if target is not None:
                nw, nh, nspp = size_for(rate * target / rays_per_sample)
                if (nw, nh, nspp) == (fw, fh, spp):
                    continue
                fw, fh, spp = nw, nh, nspp
            flat, rs = scene_cached(fw, fh)
            dt, rays = run_port(flat, fw, fh, spp, n_threads) if port else run(flat, rs, fw, fh, spp, n_threads)
            rate, rays_per_sample = rays / max(dt, 1e-6), rays / (fw * fh * spp)
        return dt, rays, fw, fh, spp

    dt, rays, fw, fh, spp = calibrated(seconds_budget, threads)
    dts, rays_s, sw, sh, sspp = calibrated(5.0, 1)
    out = {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": kind,
        "sample": f"{scene}, camera of the {w}x{h} frame at {fw}x{fh}, {spp} spp, depth 50, deterministic stream, {dt:.1f} s",
        "spp_per_s": spp * (fw * fh) / (w * h) / dt,  # in full frames
        "single_core_value": rays_s / dts / 1e6,
        "single_core_sample": f"{sw}x{sh}, {sspp} spp, {dts:.1f} s",
    }
    if kind == "reference":
        dtp, rays_p, qw, qh, qspp = calibrated(5.0, threads, port=True)
        out["port_value"] = rays_p / dtp / 1e6
        out["port_sample"] = f"CPU restatement, {threads} threads, {qw}x{qh}, {qspp} spp, {dtp:.1f} s"
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cornell_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--spp-per-step", type=int, default=0)
    ap.add_argument("--share-of", type=int, default=0,
                    help="render this GPU's share of the configuration split over that many GPUs: samples per step = the "
                         "configuration's frame (or progressive step) / SHARE_OF, e.g. 8 -> 512 / 128 / 128 / 32 spp for the "
                         "Cornell, sphere-field, 100 000-item and 4K workloads")
    ap.add_argument("--width", type=int, default=0, help="override the workload's frame width (tests)")
    ap.add_argument("--height", type=int, default=0, help="override the workload's frame height (tests)")
    ap.add_argument("--mode", default="fast", choices=["fast", "deterministic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="one render stream (the launches of consecutive steps do not overlap), reduce / resolve on it "
                         "and a synchronous copy back of the image after every step: per-kernel counters and "
                         "durations then see the trace kernel alone")
    ap.add_argument("--no-specialize", action="store_true",
                    help="keep the table-driven scan kernel for small scenes (RTX_OPTION_SPECIALIZE_SCAN = 0): A/B against "
                         "the kernel rtx_upload_scene compiles for the scene")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--single-process", action="store_true",
                    help="one process drives all --gpus devices through rtx_render_multi (the C ABI's own split: "
                         "peer copies to device 0, ordered adds) instead of one process per GPU with RCCL")
    ap.add_argument("--share-device", action="store_true",
                    help="put the ranks (or, with --single-process, the contexts) on the devices that exist, several "
                         "to a device, when fewer than --gpus are visible: a functional rehearsal of the N > 1 code on "
                         "a one-GPU box, with --backend gloo (RCCL refuses two ranks on one device); the number is "
                         "not a scaling result")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the GPU ranks (nccl = RCCL; gloo only for --share-device rehearsals)")
    ap.add_argument("--engine", default="",
                    help="test hook: dotted module name providing make_engine(scene, width, height, mode) -> object "
                         "with render(begin, count) -> (fp32 SUM frame as a CPU torch tensor, rays); the ranks then "
                         "use CPU tensors and the gloo backend, everything else (launch, sample split, pipelined "
                         "reduce, resolve, JSON line) is the code the GPU run uses")
    ap.add_argument("--dump-frame", default="", help="rank 0 saves the final fp32 SUM frame to this .npy file (tests)")
    ap.add_argument("--rank-timeout", type=float, default=900.0,
                    help="wall-clock deadline in seconds for a whole run (N > 1 ranks too): a parent that started the "
                         "ranks kills them when it passes, a rank ends itself; either way the exit code is non-zero "
                         "and stderr says where every rank was last seen")
    ap.add_argument("--collective-timeout", type=float, default=180.0,
                    help="timeout in seconds of the process group's rendezvous and collectives (N > 1)")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """`--gpus N` without a launcher: start N ranks of this script (fresh child processes, one per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), forward rank 0's JSON
    line and fail if any rank fails.  Runs before anything in this process has touched the GPU and
    never execs: the children are started with subprocess."""
    import socket
    import subprocess

    n = args.gpus
    if not args.engine:
        import torch  # device_count() does not initialise the GPU

        visible = torch.cuda.device_count()
        if visible < n and not (args.share_device and visible >= 1):
            print(f"bench.py: --gpus {n} but only {visible} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    import tempfile

    trace_dir = tempfile.mkdtemp(prefix="rtx_bench_trace_")  # every rank keeps its last position here
    deadline = time.monotonic() + args.rank_timeout
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RTX_BENCH_TRACE_FILE=os.path.join(trace_dir, f"rank{rank}.trace"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while all ranks are watched: a rank that dies takes the
    # others down with it at once (they would otherwise wait in the rendezvous or a collective
    # until its timeout)
    import threading

    captured = []
    reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    timed_out = False
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        timed_out = time.monotonic() > deadline
        if timed_out or any(c not in (None, 0) for c in codes):
            for p in procs:  # exactly the children started above
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() if c is None else c for p, c in zip(procs, codes)]
            break
        time.sleep(0.05)
    reader.join(timeout=10.0)
    if timed_out:
        print(f"bench.py: the ranks did not finish within --rank-timeout {args.rank_timeout:g} s; killed. "
              "Last position of every rank:", file=sys.stderr)
        for rank in range(n):
            print(f"  rank {rank}: {last_trace_line(os.path.join(trace_dir, f'rank{rank}.trace'))}", file=sys.stderr)
    import shutil

    shutil.rmtree(trace_dir, ignore_errors=True)
    if timed_out:
        return 124
    out = captured[0] if captured else b""
    # rank 0's JSON line goes to stdout; anything else a library printed there (gloo and RCCL
    # announce themselves on stdout) goes to stderr, so that stdout stays the one line
    for line in out.decode(errors="replace").splitlines():
        print(line, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr, flush=True)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def last_trace_line(path: str) -> str:
    try:
        with open(path) as f:
            lines = f.read().splitlines()
        return lines[-1] if lines else "(started, no position recorded)"
    except OSError:
        return "(never started)"


def trace_kernel_name(mode: str, bvh: bool) -> str:
    # fast-mode BVH frames run the kernel that decouples traversal from shading
    return "trace_bvh_kernel" if (bvh and mode == "fast") else "trace_kernel"


def run_single_process(args) -> int:
    """All GPUs from one process through rtx_render_multi (host accumulator: the PCIe copy of the
    frame is inside the timed region, unlike the per-rank path)."""
    import torch  # noqa: F401  (its HIP runtime must be the one in the process)

    from raytrace_amd import api

    n = args.gpus
    visible = api.device_count()
    if visible < 1:
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    if visible < n and not args.share_device:
        raise SystemExit(f"--gpus {n} but only {visible} GPU(s) visible (--share-device rehearses on one)")
    scene_name, w, h, spp_default = WORKLOADS[args.workload]
    w, h = args.width or w, args.height or h
    spp = args.spp_per_step or spp_default
    mode = api.RTX_MODE_FAST if args.mode == "fast" else api.RTX_MODE_DETERMINISTIC
    devices = [d if d < visible else 0 for d in range(n)]
    scene = api.HostScene(scene_name, w, h)
    ctxs = [api.Context(d) for d in devices]
    for c in ctxs:
        c.upload(scene, api.RTX_ACCEL_AUTO)
    info = ctxs[0].info()
    frame = np.zeros((h, w, 3), np.float32)

    def step(k):
        p = ctxs[0].params(w, h, k * n * spp, n * spp, mode=mode, seed=0, accumulate=True)
        return api.render_multi(ctxs, p, frame)[1]

    for k in range(args.warmup):
        step(k)
    frame[:] = 0
    t0 = time.perf_counter()
    n_rays = sum(step(args.warmup + k) for k in range(args.steps))
    elapsed = time.perf_counter() - t0
    image = api.resolve_host(frame, n * spp * args.steps)
    samples = w * h * spp * n * args.steps
    print(json.dumps({
        "metric": "Mrays/s at 1080p Cornell box (whole job)" if args.workload == "cornell_1080p"
                  else f"Mrays/s ({args.workload})",
        "value": n_rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "scene": scene_name, "width": w, "height": h,
                   "spp_per_step_per_gpu": spp, "spp_total": spp * n * args.steps, "max_depth": 50, "mode": args.mode,
                   "accel": "bvh" if info.accel == api.RTX_ACCEL_BVH else "linear", "devices": devices,
                   "parallelism": f"one process, rtx_render_multi over {n} contexts: spp-split, peer copies to "
                                  "device 0, ordered adds; the frame returns to the host every step (PCIe inside "
                                  "the timed region)"},
        "spp_per_s": spp * n * args.steps / elapsed, "rays_per_sample": n_rays / samples,
        "image_mean_rgb8": float(image.mean()),
    }), flush=True)
    for c in ctxs:
        c.close()
    return 0


def main() -> int:
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.single_process:
        return run_single_process(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args)  # before any torch.cuda / HIP call in this process
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py with --gpus equal to the "
                         "number of ranks (or without a launcher: it starts its own ranks)")

    trace_path = os.environ.get("RTX_BENCH_TRACE_FILE")
    trace_file = open(trace_path, "a", buffering=1) if trace_path else None  # line-buffered
    last_position = ["started"]

    def trace(msg):  # where a rank is: kept for the deadline's report, on stderr with RTX_BENCH_TRACE=1
        last_position[0] = msg
        if trace_file:
            trace_file.write(f"{time.strftime('%H:%M:%S')} {msg}\n")
        if os.environ.get("RTX_BENCH_TRACE"):
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    # A rank's own deadline (the launcher may be torchrun, which has none): ranks parked in a collective
    # must not run until the machine's lease ends.  os._exit ends the process without an exec and without
    # waiting for the GPU queue; the launcher then stops the other ranks.
    import threading

    def expired():
        print(f"bench.py: rank {rank} did not finish within --rank-timeout {args.rank_timeout:g} s; "
              f"last position: {last_position[0]}", file=sys.stderr, flush=True)
        os._exit(124)

    watchdog = threading.Timer(args.rank_timeout, expired)
    watchdog.daemon = True
    watchdog.start()
    trace("importing torch")

    import datetime

    import torch
    import torch.distributed as dist

    from raytrace_amd import api
    from raytrace_amd.distributed import run_pipelined_steps, step_sample_range

    pg_timeout = datetime.timedelta(seconds=args.collective_timeout)

    scene_name, w, h, spp_default = WORKLOADS[args.workload]
    w, h = args.width or w, args.height or h
    spp = args.spp_per_step or (max(1, FRAME_SPP[args.workload] // args.share_of) if args.share_of else spp_default)
    mode = api.RTX_MODE_FAST if args.mode == "fast" else api.RTX_MODE_DETERMINISTIC

    engine = None
    if args.engine:  # test hook: CPU tensors + gloo around an engine the test supplies
        import importlib

        engine = importlib.import_module(args.engine).make_engine(scene_name, w, h, args.mode)
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
        trace("process group up")
        ctx, bvh = None, False
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
        if args.share_device:
            local_rank %= torch.cuda.device_count()
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
        trace("process group up")
        ctx = api.Context(local_rank)
        scene = api.HostScene(scene_name, w, h)
        if args.no_specialize:
            ctx.set_option(api.RTX_OPTION_SPECIALIZE_SCAN, 0)
        t_upload = time.perf_counter()
        ctx.upload(scene, api.RTX_ACCEL_AUTO)
        upload_seconds = time.perf_counter() - t_upload
        scene_info = ctx.info()
        bvh = scene_info.accel == api.RTX_ACCEL_BVH

    def sync():
        if engine is None:
            torch.cuda.synchronize(dev)

    # Consecutive steps are launched on two streams in turn (unless --serial): a launch ends on its slowest
    # waves, and the next step - an independent sample range into another accumulator - fills the CUs they
    # leave idle (measured, kernels alone: Cornell box 1080p x 512 spp 103.4 -> 110.4 Grays/s, the 4K frame
    # x 32 spp 72.9 -> 79.4, the 100 000-item scene x 128 spp 8.14 -> 8.46).  Three accumulators rotate.
    overlap = engine is None and not args.serial
    step_accum = [torch.zeros((h, w, 3), dtype=torch.float32, device=dev) for _ in range(3 if overlap else 2)]  # this GPU's steps
    frame_accum = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)  # rank 0: running frame
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    if engine is None:
        # the 8-bit image of every step lands in page-locked memory: the copy back is then a DMA transfer
        # and not a series of copy kernels that share the CUs with the next step's rendering
        image_pinned = torch.empty((h, w, 3), dtype=torch.uint8).pin_memory() if rank == 0 else None
        image_host = image_pinned.numpy() if rank == 0 else None
        image_dev = torch.empty((h, w, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
        copy_stream = torch.cuda.Stream(dev) if rank == 0 else None  # the copy back of a step's image
        copied = [None]
        render_stream = torch.cuda.current_stream(dev)
        render_streams = [render_stream, torch.cuda.Stream(dev)] if overlap else [render_stream]
        # reduce (N > 1), frame addition and resolve of a step on a stream of their own, in step order, while
        # the next steps render; --serial: in line behind the step's trace kernel
        side_stream = render_stream if args.serial else torch.cuda.Stream(dev)
    else:
        render_stream = side_stream = None
        render_streams = None
    ev = {}

    total_spp_done = 0

    def render(k: int, buf):
        begin, _ = step_sample_range(k, rank, world, spp)
        timed = k >= args.warmup
        if engine is not None:
            t = time.perf_counter()
            part, n = engine.render(begin, spp)
            buf.copy_(part)
            rays.add_(int(n))
            if timed:
                ev[k] = (time.perf_counter() - t) * 1e3
            return
        p = ctx.params(w, h, begin, spp, mode=mode, seed=0, accumulate=False)
        stream = torch.cuda.current_stream(dev)  # the render stream this step was given
        if timed:
            ev[k] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[k][0].record(stream)  # on the stream the kernel is launched on
        ctx.render_device(p, buf.data_ptr(), rays.data_ptr(), stream.cuda_stream)
        if timed:
            ev[k][1].record(stream)

    def combine(k: int, buf):
        nonlocal total_spp_done
        if world > 1:
            trace(f"reduce of step {k}")
            dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
            trace(f"reduce of step {k} returned")
        total_spp_done += spp * world
        if rank == 0:
            frame_accum.add_(buf)
            if engine is not None:
                return api.resolve_host(frame_accum.numpy(), total_spp_done)
            # the 8-bit image is made on the GPU behind the frame addition; its copy into page-locked memory
            # runs on a stream of its own while the next step renders (the image buffer is not written
            # again before that copy has finished)
            if args.serial:
                # counter runs: resolve and copy back in one synchronous call.  The copy back makes the runtime
                # write the L2 back first, so that what the frame addition and the resolve left dirty there is
                # not written back inside the next trace kernel's window (and counted as its traffic)
                return ctx.resolve_device(w, h, total_spp_done, frame_accum.data_ptr(), side_stream.cuda_stream,
                                          out=image_host)
            if copied[0] is not None:
                side_stream.wait_event(copied[0])
            ctx.resolve_to_device(w, h, total_spp_done, frame_accum.data_ptr(), image_dev.data_ptr(), side_stream.cuda_stream)
            copy_stream.wait_event(side_stream.record_event())
            with torch.cuda.stream(copy_stream):
                image_pinned.copy_(image_dev, non_blocking=True)
                copied[0] = copy_stream.record_event()
            return image_host  # complete once the device has been synchronised
        return None

    trace("scene uploaded, warm-up")
    run_pipelined_steps(range(args.warmup), render, combine, step_accum, render_streams, side_stream)
    sync()
    trace("warm-up done")
    rays.zero_()
    frame_accum.zero_()  # the reported frame holds the timed steps' samples only
    total_spp_done = 0

    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    images = run_pipelined_steps(range(args.warmup, args.warmup + args.steps), render, combine, step_accum,
                                 render_streams, side_stream)
    sync()
    trace("timed steps done")
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    image = images[-1] if images else None

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rays_total = rays.clone()
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_total, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    n_rays = int(rays_total.item())
    my_rays = int(rays.item())
    kernel_ms = [v if engine is not None else v[0].elapsed_time(v[1]) for v in ev.values()]
    # The time the trace kernels of the timed steps had the GPU, per launch: from the start of the first to
    # the end of the last (HIP events on the kernels' own streams).  With one render stream that is their
    # average duration plus the gaps between them; with two, consecutive launches overlap and each one's
    # own duration also covers the time it shared the GPU with its neighbours.
    if engine is None and ev:
        first = min(ev)
        kernel_span_ms = max(ev[first][0].elapsed_time(v[1]) for v in ev.values()) / len(ev)
    else:
        kernel_span_ms = float(np.mean(kernel_ms))

    if rank == 0:
        if args.dump_frame:
            np.save(args.dump_frame, frame_accum.cpu().numpy())
        samples = w * h * spp * world * args.steps
        flops_per_ray, flops_src, flops_ref_tree = load_flops_per_ray(scene_name, bvh)
        avg_kernel_s = kernel_span_ms * 1e-3
        rays_per_launch = my_rays / args.steps
        achieved_tflops = None if flops_per_ray is None else flops_per_ray * rays_per_launch / avg_kernel_s / 1e12
        out = {
            "metric": "Mrays/s at 1080p Cornell box (whole job)" if args.workload == "cornell_1080p"
                      else f"Mrays/s ({args.workload})",
            "value": n_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",  # every GPU renders spp_per_step_per_gpu samples of the whole frame per step
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "scene": scene_name,
                "width": w, "height": h,
                "spp_per_step_per_gpu": spp,
                "spp_total": spp * world * args.steps,
                "share_of": args.share_of or None,
                "max_depth": 50,
                "mode": args.mode,
                "accel": "bvh" if bvh else "linear",
                # the scan of a small scene compiled into the kernel at upload (rtx_upload_scene; --no-specialize
                # keeps the table-driven kernel); the upload's wall time includes that compilation
                "scan_specialized": bool(scene_info.scan_specialized) if engine is None else None,
                "upload_seconds": upload_seconds if engine is None else None,
                "parallelism": (f"spp-split x{world}, reduce(sum) of fp32 accumulators to rank 0 per step, "
                                "overlapped with the next steps' rendering") if world > 1 else
                               ("one GPU: frame addition and resolve of a step in line behind its trace kernel" if args.serial or engine
                                is not None else "one GPU: consecutive steps on two streams, frame addition and resolve on a third"),
            },
            "spp_per_s": spp * world * args.steps / elapsed,
            "msamples_per_s": samples / elapsed / 1e6,
            "rays_per_sample": n_rays / samples,
            "roofline": {
                "bound": "valu_fp32",
                "achieved": achieved_tflops,
                "peak": PEAK_FP32_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": None if achieved_tflops is None else achieved_tflops / PEAK_FP32_VALU_TFLOPS,
                "traffic": load_hbm_traffic(args.workload, spp),
                "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, profiles/hbm_traffic.json)",
                "kernel": trace_kernel_name(args.mode, bvh) if engine is None else args.engine,
                "kernel_ms_avg": kernel_span_ms,
                "kernel_ms_avg_note": "start of the first to end of the last timed launch / launches (HIP events on the "
                                      "kernels' streams)" + ("; consecutive launches overlap on two streams, a launch's own "
                                                             "duration (kernel_ms_each_avg) includes the time it shares the GPU"
                                                             if overlap else ""),
                "kernel_ms_each_avg": float(np.mean(kernel_ms)),
                "kernel_ms_each_min": float(np.min(kernel_ms)),
                "launch_streams": len(render_streams) if render_streams else 1,
                "rays_per_launch": rays_per_launch,
                "kernel_grays_per_s": rays_per_launch / avg_kernel_s / 1e9,
                "flops_per_ray": flops_per_ray,
                "flops_per_ray_source": flops_src,
                # BVH scenes: what the same rays cost through the CPU restatement's own tree (median splits,
                # children visited in no particular order); not what the kernel does, kept for comparison
                "flops_per_ray_reference_tree": flops_ref_tree,
            },
            "image_mean_rgb8": float(image.mean()) if image is not None else None,
        }
        if engine is not None:
            out["engine"] = args.engine
        if not args.no_cpu_baseline and world == 1 and engine is None:
            host_cores = os.cpu_count() or 1
            threads = args.cpu_threads or min(16, host_cores)
            try:
                out["cpu_baseline"] = cpu_baseline(scene_name, w, h, threads)
                out["cpu_baseline"]["host_cores_visible"] = host_cores
            except Exception as e:  # the oracle is optional equipment, the GPU numbers are not
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)

    if world > 1:
        trace("final barrier")
        dist.barrier()
        dist.destroy_process_group()
    watchdog.cancel()
    return 0


if __name__ == "__main__":
    sys.exit(main())
