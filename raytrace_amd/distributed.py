"""Sample-range sharding across GPUs (one process per GPU, torch.distributed / RCCL).

The path shards naturally over samples: every (pixel, sample) is independent and the
random stream is keyed by the ABSOLUTE sample index, so ranks render disjoint sample
ranges of the whole frame into their own fp32 SUM accumulators and one sum-reduce to
rank 0 combines them.  There is no other exchange step.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple


def split_samples(total_spp: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous blocks [begin, begin+count) of a fixed sample budget, one per rank
    (strong split: the first total_spp % world ranks get one extra sample)."""
    if total_spp < 0 or world < 1:
        raise ValueError("bad sample budget / world size")
    base, extra = divmod(total_spp, world)
    out, begin = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((begin, n))
        begin += n
    return out


def step_sample_range(step: int, rank: int, world: int, spp_per_step: int) -> Tuple[int, int]:
    """Weak split used by progressive rendering: in step k every rank renders
    spp_per_step new samples; the union over ranks and steps is [0, steps*world*spp)."""
    if not (0 <= rank < world) or step < 0 or spp_per_step < 0:
        raise ValueError("bad step / rank")
    return (step * world + rank) * spp_per_step, spp_per_step


def render_frame(render_range: Callable[[int, int], "object"], total_spp: int, rank: int, world: int,
                 reduce_to_root: Callable[["object"], None] | None = None):
    """Render this rank's block of `total_spp` samples with `render_range(begin, count)`
    (which returns the fp32 SUM accumulator tensor of that range) and sum-reduce the
    accumulators to rank 0.  Returns the accumulator (complete on rank 0)."""
    begin, count = split_samples(total_spp, world)[rank]
    accum = render_range(begin, count)
    if world > 1:
        if reduce_to_root is None:
            import torch.distributed as dist

            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        else:
            reduce_to_root(accum)
    return accum


def run_pipelined_steps(steps: Sequence[int], render: Callable[[int, "object"], None],
                        combine: Callable[[int, "object"], "object"], buffers: Sequence["object"],
                        render_stream=None, side_stream=None) -> list:
    """Progressive steps with the exchange of step k hidden behind the rendering of the steps after it.

    `render(step, buf)` enqueues the rendering of `step` into the accumulator `buf` (on the current
    stream); `combine(step, buf)` sum-reduces `buf` to rank 0 and consumes it there (adds it to the
    running frame, resolves); its return values are collected.  The accumulators are used in turn,
    and the rendering of the next len(buffers) - 1 steps is enqueued BEFORE the current step is
    combined, so the GPU renders while an earlier step's accumulator travels over xGMI and is
    resolved.  With streams (GPU) `render` runs on `render_stream`, `combine` on `side_stream`,
    ordered by events in both directions (a buffer is not rendered into again before its last
    combine has finished); without streams (CPU tensors, gloo) the same order runs synchronously.

    `render_stream` may be a list of streams: consecutive steps then go to them in turn, so that the
    trace kernel of one step starts while the last waves of the step before it finish (a launch ends
    on its slowest waves; steps are independent, so the next one can fill the CUs they leave idle).
    That needs one more accumulator than streams.
    """
    steps = list(steps)
    cuda = render_stream is not None
    streams = list(render_stream) if isinstance(render_stream, (list, tuple)) else [render_stream]
    n_buf = len(buffers)
    if n_buf < 2 or (len(streams) > 1 and n_buf < len(streams) + 1):
        raise ValueError("two accumulators alternate; several render streams need one more accumulator than streams")
    if cuda:
        import torch
    rendered = [None] * n_buf
    combined = [None] * n_buf

    def launch(i: int) -> None:
        b = i % n_buf
        if cuda:
            rs = streams[i % len(streams)]
            if combined[b] is not None:
                rs.wait_event(combined[b])
            with torch.cuda.stream(rs):
                render(steps[i], buffers[b])
            rendered[b] = rs.record_event()
        else:
            render(steps[i], buffers[b])

    out = []
    if not steps:
        return out
    ahead = n_buf - 1  # steps whose rendering is enqueued before the current one is combined
    for i in range(min(ahead, len(steps))):
        launch(i)
    for i, step in enumerate(steps):
        if i + ahead < len(steps):
            launch(i + ahead)
        b = i % n_buf
        if cuda:
            side_stream.wait_event(rendered[b])
            with torch.cuda.stream(side_stream):
                out.append(combine(step, buffers[b]))
            combined[b] = side_stream.record_event()
        else:
            out.append(combine(step, buffers[b]))
    return out
