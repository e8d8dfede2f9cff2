"""Do consecutive launches of the trace kernel overlap when they alternate between two or three streams (the tail
of one launch filled by the head of the next)?  GPU box.  usage: python tools/overlap_probe.py scene w h spp accel [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api

name, w, h, spp, accel = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 8
ctx = api.Context(0)
ctx.upload(api.HostScene(name, w, h), accel)
acc = [torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0") for _ in range(3)]
rays = torch.zeros(1, dtype=torch.int64, device="cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
for mode in ("one stream", "two streams", "three streams", "one stream", "two streams", "three streams"):
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    rays.zero_()
    t0 = time.perf_counter()
    for k in range(n):
        n_streams = {"one stream": 1, "two streams": 2, "three streams": 3}[mode]
        s = streams[k % n_streams]
        p = ctx.params(w, h, k * spp, spp, seed=1)
        ctx.render_device(p, acc[k % n_streams].data_ptr(), rays.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name} {w}x{h} x {spp} spp, {n} launches, {mode}: {dt * 1e3 / n:.2f} ms per launch, {int(rays.item()) / dt / 1e9:.2f} Grays/s", flush=True)
