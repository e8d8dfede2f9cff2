"""Mean squared error of fast-mode and deterministic-mode renders against a high-spp ground
truth, averaged over many seeds: the ratio tells whether fast mode's estimator has the
variance of the reference's (1.0) - a correlated random stream or a biased shortcut shows here.
usage: python tools/mse_ratio.py [scene] [seeds] [spp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTX_TEXTURE", os.path.join(ROOT, "tests", "golden", "texture_64x32.bmp"))
from raytrace_amd import api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
w, h = 160, 90
ctx = api.Context(0)
ctx.upload(api.HostScene(name, w, h), api.RTX_ACCEL_LINEAR)
gt, _ = ctx.render(ctx.params(w, h, 1 << 20, 65536, mode=api.RTX_MODE_DETERMINISTIC, seed=777))
gt = np.clip(gt / 65536, 0, 1)
mse = {0: [], 1: []}
for seed in range(seeds):
    for mode in (api.RTX_MODE_FAST, api.RTX_MODE_DETERMINISTIC):
        img, _ = ctx.render(ctx.params(w, h, 0, spp, mode=mode, seed=1000 + seed))
        mse[mode].append(float(np.mean((np.clip(img / spp, 0, 1) - gt) ** 2)))
f, d = np.array(mse[0]), np.array(mse[1])
se = np.sqrt(f.var() / seeds + d.var() / seeds) / d.mean()
print(f"{name} {w}x{h} {spp} spp, {seeds} seeds: MSE fast {f.mean():.6e}  det {d.mean():.6e}  ratio {f.mean() / d.mean():.4f} +- {se:.4f}")
