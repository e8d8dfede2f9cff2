"""Spread of the equal-spp RMSE of fast and deterministic renders against a CPU ground truth
(a probe that uses the CPU oracle, hence kept with the tests; not collected by pytest)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTX_TEXTURE", os.path.join(ROOT, "tests", "golden", "texture_64x32.bmp"))
from oracle import restate  # noqa: E402
from raytrace_amd import api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "textured"
w, h, spp = 96, 54, 128
scene = api.HostScene(name, w, h)
ctx = api.Context(0)
ctx.upload(scene, api.RTX_ACCEL_LINEAR)
gt_sum, _ = ctx.render(ctx.params(w, h, 100000, 16384, mode=api.RTX_MODE_DETERMINISTIC, seed=77))
gt = np.clip(gt_sum / 16384, 0, 1)
cpu = restate.render(scene, w, h, 1000, 512, seed=5, threads=16)["sum"] / 512
print("gt(det gpu 16k) mean", gt.mean(), "cpu512 mean", np.clip(cpu, 0, 1).mean())
for seed in range(6):
    out = []
    for mode in (api.RTX_MODE_FAST, api.RTX_MODE_DETERMINISTIC):
        img, rays = ctx.render(ctx.params(w, h, 0, spp, mode=mode, seed=seed))
        img = np.clip(img / spp, 0, 1)
        out.append((float(np.sqrt(np.mean((img - gt) ** 2))), float(img.mean()), rays))
    print(seed, "fast rmse %.5f mean %.5f rays %d | det rmse %.5f mean %.5f rays %d" % (*out[0], *out[1]))
big, _ = ctx.render(ctx.params(w, h, 0, 16384, mode=api.RTX_MODE_FAST, seed=3))
big = np.clip(big / 16384, 0, 1)
print("fast 16k vs det 16k: rmse %.5f mean diff %.6f" % (np.sqrt(np.mean((big - gt) ** 2)), big.mean() - gt.mean()))
d = np.abs(big - gt).max(axis=2)
print("worst pixels", np.argwhere(d > 0.1)[:10].tolist(), float(d.max()))
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
api.write_bmp(os.path.join(out_dir, f"{name}_fast.bmp"), api.resolve_host(big * 16384, 16384))
api.write_bmp(os.path.join(out_dir, f"{name}_det.bmp"), api.resolve_host(gt_sum, 16384))
diff = np.abs(big - gt)
np.save(os.path.join(out_dir, f"{name}_diff.npy"), (big - gt).astype(np.float32))
