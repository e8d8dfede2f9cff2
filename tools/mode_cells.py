"""Fast mode against deterministic mode as estimators, cell by cell (GPU box): eight independent
sample ranges per mode, t-tests of a 16 x 9 grid of cells.  Two runs of one mode against each other
give an rms z of about 1.1; a cell whose |z| stands out marks a place where the two arithmetics
render different pictures.
usage: python tools/mode_cells.py scene width height spp [accel 0|1|2] [--map]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from raytrace_amd import api

name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
accel = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5].isdigit() else api.RTX_ACCEL_AUTO
k = 8
ctx = api.Context(0)
ctx.upload(api.HostScene(name, w, h), accel)


def chunks(mode, seed):
    frames, rays = [], 0
    for c in range(k):
        img, r = ctx.render(ctx.params(w, h, c * (spp // k), spp // k, mode=mode, seed=seed))
        frames.append(img.reshape(9, h // 9, 16, w // 16, 3).sum(axis=(1, 3, 4)).astype(np.float64))
        rays += r
    return np.stack(frames), rays


a, ra = chunks(api.RTX_MODE_FAST, 3)
b, rb = chunks(api.RTX_MODE_DETERMINISTIC, 4)
se = np.sqrt(a.var(axis=0, ddof=1) / k + b.var(axis=0, ddof=1) / k)
lit = b.mean(axis=0) > 1e-6 * b.mean(axis=0).max()
z = np.where(lit, (a.mean(axis=0) - b.mean(axis=0)) / np.maximum(se, 1e-30), 0.0)
tf, td = a.sum(axis=(1, 2)), b.sum(axis=(1, 2))
zt = (tf.mean() - td.mean()) / np.sqrt(tf.var(ddof=1) / k + td.var(ddof=1) / k)
print(f"{name} {w}x{h}x{spp} accel {ctx.info().accel}: rms z {np.sqrt(np.mean(z[lit] ** 2)):.2f}  max |z| {np.abs(z).max():.1f}  mean z {z[lit].mean():+.2f}  "
      f"frame z {zt:+.1f} ({(tf.mean() / td.mean() - 1) * 100:+.3f} %)  rays fast / det - 1 = {(ra / rb - 1) * 100:+.3f} %", flush=True)
if "--map" in sys.argv:
    np.set_printoptions(linewidth=200, precision=1, suppress=True)
    print(z)
    print("fast / det - 1, in 1e-4:")
    print(np.where(lit, (a.mean(axis=0) / np.maximum(b.mean(axis=0), 1e-30) - 1) * 1e4, 0.0))
