"""The N > 1 path on CPU: two processes, gloo backend.

The GPU kernel cannot run here, so each rank fills its accumulator with the CPU
restatement (test infrastructure standing in for the kernel — same C-level contract:
SUM over an absolute sample range).  What is under test is the sharding and the
reduce: disjoint ranges that cover the budget, streams keyed by absolute sample index,
one sum-reduce to rank 0, result independent of the world size up to fp32 summation
order."""
import os
import socket
import sys

import numpy as np
import pytest

from raytrace_amd import distributed as rd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_samples_covers_budget_without_overlap():
    for total in (0, 1, 7, 512, 4096, 4099):
        for world in (1, 2, 3, 8):
            blocks = rd.split_samples(total, world)
            assert len(blocks) == world
            assert blocks[0][0] == 0
            for (b0, n0), (b1, _) in zip(blocks, blocks[1:]):
                assert b0 + n0 == b1
            assert blocks[-1][0] + blocks[-1][1] == total
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1


def test_step_ranges_tile_the_sample_axis():
    world, spp = 4, 32
    seen = []
    for step in range(3):
        for rank in range(world):
            seen.append(rd.step_sample_range(step, rank, world, spp))
    seen.sort()
    assert seen[0][0] == 0
    for (b0, n0), (b1, _) in zip(seen, seen[1:]):
        assert b0 + n0 == b1
    with pytest.raises(ValueError):
        rd.step_sample_range(0, 4, 4, 1)


def test_pipelined_steps_order_and_buffers():
    """render(k+1) is issued before combine(k); the two accumulators alternate; a buffer is
    combined before it is rendered into again."""
    log = []
    buffers = ["A", "B"]
    out = rd.run_pipelined_steps([3, 4, 5, 6], lambda k, b: log.append(("render", k, b)),
                                 lambda k, b: log.append(("combine", k, b)) or k * 10, buffers)
    assert out == [30, 40, 50, 60]
    assert log == [("render", 3, "A"), ("render", 4, "B"), ("combine", 3, "A"), ("render", 5, "A"),
                   ("combine", 4, "B"), ("render", 6, "B"), ("combine", 5, "A"), ("combine", 6, "B")]
    assert rd.run_pipelined_steps([], None, None, buffers) == []
    with pytest.raises(ValueError):
        rd.run_pipelined_steps([0], None, None, ["only one"])
    # a ring of three accumulators: two steps ahead of the combine, each buffer combined before it is reused
    log.clear()
    out = rd.run_pipelined_steps(range(5), lambda k, b: log.append(("render", k, b)),
                                 lambda k, b: log.append(("combine", k, b)) or k, ["A", "B", "C"])
    assert out == [0, 1, 2, 3, 4]
    assert log == [("render", 0, "A"), ("render", 1, "B"), ("render", 2, "C"), ("combine", 0, "A"), ("render", 3, "A"),
                   ("combine", 1, "B"), ("render", 4, "B"), ("combine", 2, "C"), ("combine", 3, "A"), ("combine", 4, "B")]
    with pytest.raises(ValueError):  # two render streams want three accumulators
        rd.run_pipelined_steps([0], None, None, ["A", "B"], render_stream=[object(), object()], side_stream=object())


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import restate
    from raytrace_amd import api, distributed

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, total = 40, 24, 9
    scene = api.HostScene("cornell", w, h)

    def render_range(begin, count):
        out = restate.render(scene, w, h, begin, count, seed=21, threads=2)
        return torch.from_numpy(out["sum"].copy())

    acc = distributed.render_frame(render_range, total, rank, world)
    # progressive (weak) variant: 2 steps of 2 spp per rank, reduced every step
    # (the loop bench.py runs: the next step is rendered before the current one is reduced)
    frame = torch.zeros((h, w, 3))
    buffers = [torch.zeros((h, w, 3)), torch.zeros((h, w, 3))]

    def render(step, buf):
        b, n = distributed.step_sample_range(step, rank, world, 2)
        buf.copy_(render_range(b, n))

    def combine(step, buf):
        dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            frame.add_(buf)
        return step

    assert distributed.run_pipelined_steps(range(2), render, combine, buffers) == [0, 1]
    if rank == 0:
        np.savez(out_path, strong=acc.numpy(), weak=frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_to_the_single_process_frame(tmp_path):
    import torch.multiprocessing as mp

    from oracle import restate
    from raytrace_amd import api

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    w, h = 40, 24
    scene = api.HostScene("cornell", w, h)
    whole9 = restate.render(scene, w, h, 0, 9, seed=21, threads=4)["sum"]
    whole8 = restate.render(scene, w, h, 0, 8, seed=21, threads=4)["sum"]
    # same samples, summed in a different association: fp32 summation-order tolerance
    assert np.allclose(got["strong"], whole9, rtol=1e-5, atol=1e-6)
    assert np.allclose(got["weak"], whole8, rtol=1e-5, atol=1e-6)
    assert got["strong"].sum() > 0
