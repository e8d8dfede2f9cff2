"""One-rank RCCL self-test of the pipelined step loop (GPU box): the process group is created
with world size 1, so dist.reduce is degenerate, but the stream / event choreography that
bench.py uses for N > 1 (reduce + resolve on a side stream while the next step renders) runs
against the real RCCL backend.  usage: python tools/dist_selftest.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from raytrace_amd import api
from raytrace_amd.distributed import run_pipelined_steps, step_sample_range

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
w, h, spp = 320, 180, 8
ctx = api.Context(0)
ctx.upload(api.HostScene("cornell", w, h), api.RTX_ACCEL_AUTO)
bufs = [torch.zeros((h, w, 3), device=dev) for _ in range(2)]
frame = torch.zeros((h, w, 3), device=dev)
main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)


def render(k, buf):
    b, n = step_sample_range(k, 0, 1, spp)
    ctx.render_device(ctx.params(w, h, b, n, seed=3), buf.data_ptr(), 0, main.cuda_stream)


def combine(k, buf):
    dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
    frame.add_(buf)
    return ctx.resolve_device(w, h, (k + 1) * spp, frame.data_ptr(), side.cuda_stream)


imgs = run_pipelined_steps(range(6), render, combine, bufs, main, side)
torch.cuda.synchronize()
whole, _ = ctx.render(ctx.params(w, h, 0, 6 * spp, seed=3))
err = float((frame.cpu() - torch.from_numpy(whole)).abs().max())
print("pipelined 6 x 8 spp vs one 48-spp launch: max abs diff", err, "mean", float(frame.mean()) / (6 * spp))
assert err < 1e-3 * max(1.0, float(torch.from_numpy(whole).abs().max())), err
dist.barrier()
dist.destroy_process_group()
print("ok")
