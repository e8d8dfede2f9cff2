"""Render engine for bench.py's `--engine` test hook: the CPU restatement stands in for the HIP
kernel (same contract: fp32 SUM frame of an absolute sample range, row 0 = top, plus the ray
count), so that bench.py's own launch path, sample split, pipelined reduce and JSON line can run
with world size 2 on a machine without a GPU.  Test infrastructure: lives under tests/ because it
uses the oracle."""
import os
import time

import torch

from oracle import restate
from raytrace_amd import api


class _Engine:
    def __init__(self, scene: str, width: int, height: int, mode: str):
        self.w, self.h = width, height
        self.scene = api.HostScene(scene, width, height)

    def render(self, begin: int, count: int):
        if os.environ.get("BENCH_ENGINE_HANG_RANK") == os.environ.get("RANK", "0"):
            while True:  # a rank that never comes back from its render step (tests/test_bench_launch.py)
                time.sleep(1.0)
        out = restate.render(self.scene, self.w, self.h, begin, count, seed=0, threads=2)
        return torch.from_numpy(out["sum"].copy()), int(out["rays"])


def make_engine(scene: str, width: int, height: int, mode: str):
    if os.environ.get("BENCH_ENGINE_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit(3)  # a rank that dies before the rendezvous (tests/test_bench_launch.py)
    return _Engine(scene, width, height, mode)
