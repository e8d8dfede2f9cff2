"""Progressive render of the 4K configuration (Cornell + extra glass / metal, 3840x2160,
steps of 256 spp): per-step throughput, RMSE against an independent ground truth as a
function of the accumulated spp, and preview images at a few steps.

    python tools/progressive.py [--steps 64] [--spp-per-step 256] [--width 3840] [--height 2160]
                                [--out gpurun_out/progressive]

The ground truth is a separate render (other seed, deterministic mode) of the same scene at
1/4 resolution per axis and 16x the per-pixel samples of the final frame's 4x4 pixel blocks;
RMSE is taken on the 4x4 box-filtered frame, so it measures the estimator, not the pixel grid.
Beside it, where the frame has the fixture's aspect ratio, the RMSE against the REFERENCE tracer's
own image of the scene (tests/golden/ground_truth_ref_<scene>_240x135.npz: 240 x 135 pixels x 4 096
spp, rendered by the unmodified reference sources on the CPU): a ground truth that shares nothing
with the GPU code, at the price of a higher noise floor (the summary states both floors).
One JSON line per step, then a summary line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell_extra")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--spp-per-step", type=int, default=256)
    ap.add_argument("--gt-spp", type=int, default=262144,
                    help="per quarter-resolution pixel; the final frame has 16 x steps x spp-per-step samples per "
                         "such pixel, so the default equals the final frame's count: the RMSE floor it leaves is "
                         "stated in the summary")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "progressive"))
    ap.add_argument("--previews", default="1,4,16,64")
    ap.add_argument("--checkpoint", default="", help="npz file: resume from it if present, rewrite it as the run goes")
    ap.add_argument("--checkpoint-every", type=int, default=8, help="steps between two checkpoints")
    args = ap.parse_args()

    import torch

    from raytrace_amd import api

    w, h = args.width, args.height
    assert w % 4 == 0 and h % 4 == 0
    os.makedirs(args.out, exist_ok=True)
    ctx = api.Context(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)

    # ground truth at quarter resolution (same camera: the aspect ratio is unchanged)
    gw, gh = w // 4, h // 4
    ctx.upload(api.HostScene(args.scene, gw, gh), api.RTX_ACCEL_AUTO)
    gt = torch.zeros((gh, gw, 3), dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    chunk = 16384  # launches of a second or two each
    for begin in range(0, args.gt_spp, chunk):
        ctx.render_device(ctx.params(gw, gh, (1 << 24) + begin, min(chunk, args.gt_spp - begin),
                                     mode=api.RTX_MODE_DETERMINISTIC, seed=4242, accumulate=True),
                          gt.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize(dev)
    gt = torch.clamp(gt / args.gt_spp, 0.0, 1.0)  # the 8-bit image clamps too (TonemapFilter.h:8-10)
    print(json.dumps({"ground_truth": f"{gw}x{gh}, {args.gt_spp} spp, deterministic, seed 4242",
                      "seconds": time.perf_counter() - t0}), flush=True)

    # the reference tracer's own mean image of the scene, if the test fixtures hold one of this aspect ratio
    ref_gt = None
    ref_path = os.path.join(ROOT, "tests", "golden", f"ground_truth_ref_{args.scene}_240x135.npz")
    if os.path.exists(ref_path) and w % 240 == 0 and h % 135 == 0 and w // 240 == h // 135:
        g = np.load(ref_path)
        ref_gt = torch.clamp(torch.from_numpy(g["mean"].astype(np.float32)).to(dev), 0.0, 1.0)
        ref_block = w // 240
        print(json.dumps({"reference_ground_truth": os.path.relpath(ref_path, ROOT), "spp": int(g["spp"]),
                          "pixels": "240x135", "block": ref_block}), flush=True)

    from raytrace_amd.progressive import ProgressiveFrame

    ctx.upload(api.HostScene(args.scene, w, h), api.RTX_ACCEL_AUTO)
    prog = ProgressiveFrame(ctx, w, h, args.spp_per_step, seed=0, mode=api.RTX_MODE_FAST)
    frame, rays = prog.frame, prog.rays
    first_step = 0
    if args.checkpoint and os.path.exists(args.checkpoint):
        first_step = prog.load(args.checkpoint) // args.spp_per_step
        print(json.dumps({"resumed_from": args.checkpoint, "steps_done": first_step}), flush=True)
    previews = {int(x) for x in args.previews.split(",") if x}
    rows = []
    for k in range(first_step, args.steps):
        rays_before = int(rays.item())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        spp = prog.step(stream)
        b.record(stream)
        # 4x4 box filter of the running mean (= the quarter-resolution pixel integral), clamped
        # like the ground truth
        small = (frame / spp).reshape(gh, 4, gw, 4, 3).mean(dim=(1, 3))
        mean = torch.clamp(small, 0.0, 1.0)
        rmse = float(torch.sqrt(torch.mean((mean - gt) ** 2)).item())
        rmse_ref = None
        if ref_gt is not None:
            coarse = torch.clamp((frame / spp).reshape(135, ref_block, 240, ref_block, 3).mean(dim=(1, 3)), 0.0, 1.0)
            rmse_ref = float(torch.sqrt(torch.mean((coarse - ref_gt) ** 2)).item())
        torch.cuda.synchronize(dev)
        ms = a.elapsed_time(b)
        r = int(rays.item()) - rays_before
        if args.checkpoint and ((k + 1) % args.checkpoint_every == 0 or k + 1 == args.steps):
            prog.save(args.checkpoint)
        row = {"step": k + 1, "spp": spp, "kernel_ms": ms, "grays_per_s": r / ms / 1e6, "rmse": rmse}
        if rmse_ref is not None:
            row["rmse_vs_reference"] = rmse_ref
        rows.append(row)
        print(json.dumps(row), flush=True)
        if (k + 1) in previews:
            # quarter-resolution preview through the reference's filter chain (Image.h:30-39)
            img = api.resolve_host(small.cpu().numpy(), 1)
            api.write_bmp(os.path.join(args.out, f"preview_{spp:06d}spp.bmp"), img)
    # RMSE of a Monte Carlo mean falls as spp^-1/2 until the ground truth's own noise floor
    if not rows:  # the checkpoint already held every step
        print(json.dumps({"summary": True, "scene": args.scene, "steps": args.steps, "steps_run": 0}), flush=True)
        return
    first, last = rows[0], rows[-1]
    slope = np.log(last["rmse"] / first["rmse"]) / np.log(last["spp"] / first["spp"]) if len(rows) > 1 else float("nan")
    # model: rmse^2(n) = A / n + F, F = the ground truth's own variance (it does not shrink with n).
    # A and F from the first and the last step; the slope of the part that does shrink is then
    # checked at the middle of the run (-0.5 for a Monte Carlo mean).
    floor = slope_mc = float("nan")
    if len(rows) > 2:
        n1, n2 = first["spp"], last["spp"]
        A = (first["rmse"] ** 2 - last["rmse"] ** 2) / (1.0 / n1 - 1.0 / n2)
        F = first["rmse"] ** 2 - A / n1
        floor = float(np.sqrt(max(F, 0.0)))
        mid = rows[len(rows) // 8]
        shrinking = mid["rmse"] ** 2 - F
        if shrinking > 0 and first["rmse"] ** 2 - F > 0 and mid["spp"] != n1:
            slope_mc = float(0.5 * np.log(shrinking / (first["rmse"] ** 2 - F)) / np.log(mid["spp"] / n1))
    ref_summary = {}
    if ref_gt is not None and len(rows) > 2:
        # the same model against the reference's image: its floor is the reference's own noise at 4 096 spp
        # (and whatever the two tracers' expectations differ by)
        r1, r2 = first["rmse_vs_reference"], last["rmse_vs_reference"]
        A = (r1 ** 2 - r2 ** 2) / (1.0 / first["spp"] - 1.0 / last["spp"])
        F = r1 ** 2 - A / first["spp"]
        ref_summary = {"rmse_vs_reference_first": r1, "rmse_vs_reference_last": r2,
                       "reference_noise_floor_rmse": float(np.sqrt(max(F, 0.0)))}
    print(json.dumps({"summary": True, "scene": args.scene, "width": w, "height": h, "steps": args.steps, **ref_summary,
                      "spp_per_step": args.spp_per_step, "mean_grays_per_s": float(np.mean([r["grays_per_s"] for r in rows])),
                      "rmse_first": first["rmse"], "rmse_last": last["rmse"], "loglog_slope": float(slope),
                      "ground_truth_noise_floor_rmse": floor, "loglog_slope_above_floor": slope_mc,
                      "ground_truth_spp_per_block": args.gt_spp,
                      "final_frame_samples_per_block": 16 * last["spp"]}), flush=True)


if __name__ == "__main__":
    main()
