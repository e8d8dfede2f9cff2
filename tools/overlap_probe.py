"""Do consecutive render launches on two streams backfill each other's tail?  (GPU box)
usage: python tools/overlap_probe.py [spp] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h = 1920, 1080
ctx = api.Context(0)
ctx.upload(api.HostScene("cornell", w, h), api.RTX_ACCEL_AUTO)
bufs = [torch.zeros((h, w, 3), device="cuda:0") for _ in range(2)]
rays = torch.zeros(1, dtype=torch.int64, device="cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(two_streams):
    torch.cuda.synchronize()
    rays.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(steps):
        s = streams[k % 2] if two_streams else streams[0]
        ctx.render_device(ctx.params(w, h, k * spp, spp, seed=1), bufs[k % 2].data_ptr(), rays.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return dt, int(rays.item())


run(False)
for mode in (False, True, False, True):
    dt, r = run(mode)
    print(f"two_streams={mode}: {dt * 1e3 / steps:.2f} ms/step  {r / dt / 1e9:.2f} Grays/s", flush=True)
