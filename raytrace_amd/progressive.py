"""Progressive accumulation with a checkpoint: the running fp32 SUM frame plus the cursor on
the sample axis.

The reference writes its image once, at the very end (Scene.cpp:155), and has nothing to resume
from.  Here a frame grows in steps of `spp_per_step` samples; every sample's random stream is
keyed by its ABSOLUTE index, so a frame resumed from a checkpoint continues with exactly the
samples the uninterrupted run would have rendered next, and ends bit-identical to it.
"""
from __future__ import annotations

import os

import numpy as np


class ProgressiveFrame:
    """Running SUM accumulator [h, w, 3] (a torch tensor on the render device) and the number of
    samples per pixel already in it."""

    def __init__(self, ctx, width: int, height: int, spp_per_step: int, seed: int = 0, mode: int = 0,
                 max_depth: int = 50, device=None):
        import torch

        self.ctx, self.w, self.h = ctx, width, height
        self.spp_per_step, self.seed, self.mode, self.max_depth = spp_per_step, seed, mode, max_depth
        self.device = device if device is not None else torch.device("cuda", ctx.device)
        self.frame = torch.zeros((height, width, 3), dtype=torch.float32, device=self.device)
        self.rays = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.spp_done = 0

    def step(self, stream=None) -> int:
        """Add the next `spp_per_step` samples to the frame (enqueued on `stream`, default the
        current one).  Returns the samples per pixel accumulated so far."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        p = self.ctx.params(self.w, self.h, self.spp_done, self.spp_per_step, mode=self.mode, seed=self.seed,
                            max_depth=self.max_depth, accumulate=True)
        self.ctx.render_device(p, self.frame.data_ptr(), self.rays.data_ptr(), s.cuda_stream)
        self.spp_done += self.spp_per_step
        return self.spp_done

    def mean(self):
        return self.frame / max(self.spp_done, 1)

    # -- checkpoint ---------------------------------------------------------------------------
    def save(self, path: str) -> None:
        """Write the accumulator and the cursor; the file replaces `path` atomically."""
        import torch

        torch.cuda.synchronize(self.device)
        tmp = path + ".tmp.npz"
        np.savez(tmp, frame=self.frame.cpu().numpy(), spp_done=np.int64(self.spp_done), rays=self.rays.cpu().numpy(),
                 width=np.int64(self.w), height=np.int64(self.h), seed=np.uint64(self.seed), mode=np.int64(self.mode),
                 max_depth=np.int64(self.max_depth))
        os.replace(tmp, path)

    def load(self, path: str) -> int:
        """Resume from a checkpoint written by save().  The frame must be the same job (size,
        seed, mode, depth): anything else would splice two different estimators."""
        import torch

        with np.load(path) as z:
            same = (int(z["width"]) == self.w and int(z["height"]) == self.h and int(z["seed"]) == self.seed
                    and int(z["mode"]) == self.mode and int(z["max_depth"]) == self.max_depth)
            if not same:
                raise ValueError(f"{path} is a checkpoint of another job")
            self.frame.copy_(torch.from_numpy(z["frame"]))
            self.rays.copy_(torch.from_numpy(z["rays"]))
            self.spp_done = int(z["spp_done"])
        return self.spp_done
