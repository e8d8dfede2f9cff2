#!/usr/bin/env python3
"""Throughput benchmark of the path-trace hot path.

    python bench.py --gpus N --steps K --warmup W

One *step* = one render pass of `--spp-per-step` samples per pixel per GPU over
the whole frame (default workload: the reference Cornell box at 1920x1080,
512 spp per GPU per step — with 8 GPUs one step is the 4096-spp frame of the
headline configuration, with one GPU it is one GPU's share of it), followed for
N > 1 by the RCCL sum-reduce of the fp32 accumulators to rank 0 and, on rank 0,
the resolve to 8 bits; reduce and resolve of a step run on a second stream while the next
step renders (two accumulators alternate).  Ranks render disjoint absolute sample ranges (weak
scaling: per-GPU work is fixed); the random stream is keyed by the absolute
sample index, so the image does not depend on N beyond fp32 summation order.

The scene is resident in HBM before the timed region; the accumulator never
leaves the device inside it (only the 8-bit resolve is copied to the host).

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
    "random_field_1080p": ("random_field", 1920, 1080, 256),  # 4 steps = the 1024-spp frame
    "bvh_stress_1080p": ("bvh_stress", 1920, 1080, 128),      # 8 GPUs x 1 step = the 1024-spp frame
    "cornell_extra_4k": ("cornell_extra", 3840, 2160, 256),
    "small_400x225": ("small", 400, 225, 100),
}


def load_flops_per_ray(scene: str, bvh: bool):
    """Algorithmic fp32 operations per ray, counted by the op-counting build of the CPU
    restatement (tests/golden/flops_per_ray.json, made by tests/golden/make_golden.py): the
    reference's O(N) scan for scenes rendered by the scan, the restatement's own hierarchy
    traversal for scenes rendered through the BVH.  (None, reason) when no count exists."""
    path = os.path.join(ROOT, "tests", "golden", "flops_per_ray.json")
    key = scene + ":bvh" if bvh else scene
    try:
        with open(path) as f:
            table = json.load(f)
        return float(table[key]["flops_per_ray"]), "counted (BVH traversal of the CPU restatement)" if bvh else "counted"
    except Exception:
        return None, f"no count for {key}"


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
    the restatement is the baseline (kind "port")."""
    from oracle import ref, restate
    from raytrace_amd import api

    flat = api.HostScene(scene, w, h)

    def run_port(spp, n_threads):
        t = time.perf_counter()
        out = restate.render(flat, w, h, 0, spp, seed=0, threads=n_threads)
        return time.perf_counter() - t, out["rays"]

    if ref.available():
        kind = "reference"
        rs = ref.RefScene(scene, w, h)

        def run(spp, n_threads=threads):
            t = time.perf_counter()
            out = rs.render(0, spp, seed=0, threads=n_threads)
            return time.perf_counter() - t, out["rays"]
    else:
        kind = "port"

        def run(spp, n_threads=threads):
            return run_port(spp, n_threads)

    dt1, rays1 = run(1)
    spp = int(max(1, min(64, seconds_budget / max(dt1, 1e-3))))
    dt, rays = (dt1, rays1) if spp == 1 else run(spp)
    dts, rays_s = run(1, 1)  # one sample per pixel on one core
    out = {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": kind,
        "sample": f"{scene} {w}x{h}, {spp} spp, depth 50, deterministic stream, {dt:.1f} s",
        "spp_per_s": spp / dt,
        "single_core_value": rays_s / dts / 1e6,
        "single_core_sample": f"1 spp, {dts:.1f} s",
    }
    if kind == "reference":
        dtp1, _ = run_port(1, threads)
        spp_p = int(max(1, min(64, 5.0 / max(dtp1, 1e-3))))
        dtp, rays_p = run_port(spp_p, threads)
        out["port_value"] = rays_p / dtp / 1e6
        out["port_sample"] = f"CPU restatement, {threads} threads, {spp_p} spp, {dtp:.1f} s"
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--spp-per-step", type=int, default=0)
    ap.add_argument("--mode", default="fast", choices=["fast", "deterministic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="reduce / resolve on the render stream (no overlap): per-kernel counters then see the "
                         "trace kernel alone")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytrace_amd import api
    from raytrace_amd.distributed import run_pipelined_steps, step_sample_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    scene_name, w, h, spp_default = WORKLOADS[args.workload]
    spp = args.spp_per_step or spp_default
    mode = api.RTX_MODE_FAST if args.mode == "fast" else api.RTX_MODE_DETERMINISTIC

    ctx = api.Context(local_rank)
    scene = api.HostScene(scene_name, w, h)
    ctx.upload(scene, api.RTX_ACCEL_AUTO)
    info = ctx.info()

    step_accum = [torch.zeros((h, w, 3), dtype=torch.float32, device=dev) for _ in range(2)]  # this GPU's steps
    frame_accum = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)  # rank 0: running frame
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    render_stream = torch.cuda.current_stream(dev)
    # reduce + resolve of step k while step k+1 renders
    side_stream = render_stream if args.serial else torch.cuda.Stream(dev)
    ev = {}

    total_spp_done = 0

    def render(k: int, buf):
        begin, _ = step_sample_range(k, rank, world, spp)
        p = ctx.params(w, h, begin, spp, mode=mode, seed=0, accumulate=False)
        timed = k >= args.warmup
        if timed:
            ev[k] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[k][0].record(render_stream)  # on the stream the kernel is launched on
        ctx.render_device(p, buf.data_ptr(), rays.data_ptr(), render_stream.cuda_stream)
        if timed:
            ev[k][1].record(render_stream)

    def combine(k: int, buf):
        nonlocal total_spp_done
        if world > 1:
            dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
        total_spp_done += spp * world
        if rank == 0:
            frame_accum.add_(buf)
            return ctx.resolve_device(w, h, total_spp_done, frame_accum.data_ptr(), side_stream.cuda_stream)
        return None

    run_pipelined_steps(range(args.warmup), render, combine, step_accum, render_stream, side_stream)
    torch.cuda.synchronize(dev)
    rays.zero_()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    images = run_pipelined_steps(range(args.warmup, args.warmup + args.steps), render, combine, step_accum,
                                 render_stream, side_stream)
    torch.cuda.synchronize(dev)
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
    kernel_ms = [a.elapsed_time(b) for a, b in ev.values()]

    if rank == 0:
        samples = w * h * spp * world * args.steps
        flops_per_ray, flops_src = load_flops_per_ray(scene_name, info.accel == api.RTX_ACCEL_BVH)
        avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "scene": scene_name,
                "width": w, "height": h,
                "spp_per_step_per_gpu": spp,
                "spp_total": spp * world * args.steps,
                "max_depth": 50,
                "mode": args.mode,
                "accel": "bvh" if info.accel == api.RTX_ACCEL_BVH else "linear",
                "parallelism": f"spp-split x{world}, reduce(sum) of fp32 accumulators to rank 0 per step, "
                               "overlapped with the next step's rendering",
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
                "kernel": "trace_kernel",
                "kernel_ms_avg": float(np.mean(kernel_ms)),
                "kernel_ms_min": float(np.min(kernel_ms)),
                "rays_per_launch": rays_per_launch,
                "kernel_grays_per_s": rays_per_launch / avg_kernel_s / 1e9,
                "flops_per_ray": flops_per_ray,
                "flops_per_ray_source": flops_src,
            },
            "image_mean_rgb8": float(image.mean()) if image is not None else None,
        }
        if not args.no_cpu_baseline and world == 1:
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            try:
                out["cpu_baseline"] = cpu_baseline(scene_name, w, h, threads)
            except Exception as e:  # the oracle is optional equipment, the GPU numbers are not
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
