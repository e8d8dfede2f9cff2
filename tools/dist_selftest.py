"""RCCL self-test of the pipelined step loop on ONE GPU (GPU box).

    python tools/dist_selftest.py [world]

world = 1 (default): the process group has one rank, so dist.reduce is degenerate, but the stream /
event choreography bench.py uses for N > 1 (reduce + resolve on a side stream while the next step
renders) runs against the real RCCL backend.  world = 2: two processes share GPU 0 (RCCL may
refuse two ranks on one device; the script then says so and exits 3), which exercises the real
sum-reduce to rank 0 as well.  The reduced frame is compared with one launch over the same
absolute sample range."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    import torch
    import torch.distributed as dist

    from raytrace_amd import api
    from raytrace_amd.distributed import run_pipelined_steps, step_sample_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        probe = torch.ones(4, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as e:  # two ranks on one device
        print(f"rank {rank}: RCCL refused this layout: {type(e).__name__}: {str(e)[:200]}", flush=True)
        os._exit(3)
    w, h, spp, steps = 320, 180, 8, 6
    ctx = api.Context(0)
    ctx.upload(api.HostScene("cornell", w, h), api.RTX_ACCEL_AUTO)
    bufs = [torch.zeros((h, w, 3), device=dev) for _ in range(2)]
    frame = torch.zeros((h, w, 3), device=dev)
    main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)

    def render(k, buf):
        b, n = step_sample_range(k, rank, world, spp)
        ctx.render_device(ctx.params(w, h, b, n, seed=3), buf.data_ptr(), 0, main.cuda_stream)

    def combine(k, buf):
        dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            frame.add_(buf)
            return ctx.resolve_device(w, h, (k + 1) * spp * world, frame.data_ptr(), side.cuda_stream)
        return None

    run_pipelined_steps(range(steps), render, combine, bufs, main, side)
    torch.cuda.synchronize()
    if rank == 0:
        total = steps * spp * world
        whole, _ = ctx.render(ctx.params(w, h, 0, total, seed=3))
        err = float((frame.cpu() - torch.from_numpy(whole)).abs().max())
        print(f"world {world}: pipelined {steps} x {spp} spp x {world} ranks vs one {total}-spp launch: max abs diff {err}, "
              f"mean {float(frame.mean()) / total}", flush=True)
        assert err < 1e-3 * max(1.0, float(torch.from_numpy(whole).abs().max())), err
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("ok", flush=True)


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    if world == 1:
        worker(0, 1, 29533)
    else:
        import torch.multiprocessing as mp

        mp.spawn(worker, args=(world, 29534), nprocs=world, join=True)
