"""Fast-mode frames of a scene with and without the scan compiled for it at upload: upload times,
equality of the two frames, kernel time of each (HIP events, alternating).

usage: python tools/special_ab.py [scene [width height spp [reps]]]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from raytrace_amd import api  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    w, h, spp = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1920, 1080, 512)
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    scene = api.HostScene(name, w, h)
    ctxs = {}
    for label, on in (("generic", 0), ("special", 1)):
        ctx = api.Context(0)
        ctx.set_option(api.RTX_OPTION_SPECIALIZE_SCAN, on)
        t = time.time()
        ctx.upload(scene, api.RTX_ACCEL_LINEAR)
        info = ctx.info()
        print(f"{label}: upload {time.time() - t:.3f} s, specialized {info.scan_specialized}, compile {info.specialize_seconds:.3f} s, note '{ctx.note()}'", flush=True)
        ctxs[label] = ctx
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev)
    out = {k: torch.zeros((h, w, 3), dtype=torch.float32, device=dev) for k in ctxs}
    rays = {k: torch.zeros(1, dtype=torch.int64, device=dev) for k in ctxs}
    small = api.Context.params(min(w, 256), min(h, 256), 0, 16)
    for k, ctx in ctxs.items():
        ctx.render_device(small, out[k].data_ptr(), rays[k].data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    p = api.Context.params(w, h, 0, spp)
    times = {k: [] for k in ctxs}
    for _ in range(reps):
        for k, ctx in ctxs.items():
            out[k].zero_()
            rays[k].zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            ctx.render_device(p, out[k].data_ptr(), rays[k].data_ptr(), stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1))
    a, b = out["generic"].cpu().numpy(), out["special"].cpu().numpy()
    ra, rb = int(rays["generic"].item()), int(rays["special"].item())
    differ = int((a != b).any(axis=2).sum())
    rel = float(np.abs(a - b).max() / max(np.abs(a).max(), 1e-30))
    print(f"rays generic {ra} special {rb}; pixels that differ {differ} of {w * h}; max |diff| / max {rel:.3g}; means {a.mean():.6f} {b.mean():.6f}")
    for k in ctxs:
        ms = min(times[k])
        n = ra if k == "generic" else rb
        print(f"{k}: min {ms:.3f} ms  avg {sum(times[k]) / len(times[k]):.3f} ms  {n / ms / 1e6:.2f} Grays/s")
    print(f"special / generic: {min(times['generic']) / min(times['special']):.4f}")


if __name__ == "__main__":
    main()
