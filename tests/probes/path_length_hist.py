"""Path-length histograms of deterministic vs fast mode on the GPU (development probe).
usage: python tests/probes/path_length_hist.py scene w h spp accel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytrace_amd import api

name, w, h, spp, accel = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ctx = api.Context(0)
ctx.upload(api.HostScene(name, w, h), accel)
def hist(mode, seed):
    rgb, seg = ctx.render_samples(ctx.params(w, h, 0, spp, mode=mode, seed=seed))
    return np.bincount(seg.ravel(), minlength=52)[:52].astype(np.int64), float(np.nanmean(rgb))
d1, m1 = hist(1, 1); d2, m2 = hist(1, 2); f1, m3 = hist(0, 3); f2, m4 = hist(0, 4)
n = d1.sum()
print("samples", n, "means det", m1, m2, "fast", m3, m4)
print("rays/sample det", (d1 * np.arange(52)).sum() / n, (d2 * np.arange(52)).sum() / n, "fast", (f1 * np.arange(52)).sum() / n, (f2 * np.arange(52)).sum() / n)
print("len   det1     det2     fast1    fast2   (fast-det)/sqrt(det)")
for k in range(1, 52):
    if d1[k] + f1[k] < 100: continue
    print(f"{k:3d} {d1[k]:9d} {d2[k]:9d} {f1[k]:9d} {f2[k]:9d}   {((f1[k]+f2[k]) - (d1[k]+d2[k])) / np.sqrt(d1[k]+d2[k]+1):8.1f}   det-det {(d1[k]-d2[k]) / np.sqrt(d1[k]+d2[k]+1):6.1f}")
