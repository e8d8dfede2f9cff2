"""Time rtx_render_device for ad-hoc configurations (GPU box).
usage: python tools/gpu_time.py scene w h spp mode(0|1) max_depth accel(0|1|2) [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api


def run(ctx, name, w, h, spp, mode, max_depth, accel, reps=3):
    sc = api.HostScene(name, w, h)
    ctx.upload(sc, accel)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    rays = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    s = torch.cuda.current_stream()
    p = ctx.params(w, h, 0, spp, mode=mode, seed=1, max_depth=max_depth)
    ctx.render_device(p, acc.data_ptr(), rays.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        rays.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        ctx.render_device(p, acc.data_ptr(), rays.data_ptr(), s.cuda_stream)
        b.record(s)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    r = int(rays.item())
    print(f"{name} {w}x{h} spp={spp} mode={mode} depth={max_depth} accel={accel}: {best:.2f} ms  rays={r} "
          f"({r / (w * h * spp):.3f}/sample)  {r / best / 1e6:.2f} Grays/s  mean={acc.mean().item() / spp:.4f}", flush=True)


if __name__ == "__main__":
    ctx = api.Context(0)
    if len(sys.argv) > 1:
        a = sys.argv
        run(ctx, a[1], int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7]), int(a[8]) if len(a) > 8 else 3)
    else:
        for mode in (0, 1):
            run(ctx, "cornell", 1920, 1080, 256, mode, 50, 1)
            run(ctx, "cornell", 1920, 1080, 256, mode, 0, 1)
            run(ctx, "cornell", 1920, 1080, 256, mode, 1, 1)
        run(ctx, "cornell_classic", 1920, 1080, 128, 0, 50, 1)
        run(ctx, "cornell_extra", 1920, 1080, 128, 0, 50, 1)
        run(ctx, "small", 1920, 1080, 128, 0, 50, 1)
        run(ctx, "random_field", 1920, 1080, 16, 0, 50, 2)
        run(ctx, "bvh_stress", 1920, 1080, 4, 0, 50, 2)
