"""Nearest hits of random rays in one scene through the four combinations deterministic / fast x
scan / tree (GPU box): how often the fast mode's winner differs from the deterministic one's.
usage: python tools/hit_modes.py scene [n_rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytrace_amd import api

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
rng = np.random.default_rng(3)
o = rng.uniform(0, 100, (n, 3)).astype(np.float32)
o[:, 1] = rng.uniform(0.1, 25, n)
d = rng.normal(size=(n, 3)).astype(np.float32)
ctx = api.Context(0)
scene = api.HostScene(name, 64, 64)
out = {}
for accel, an in ((api.RTX_ACCEL_LINEAR, "scan"), (api.RTX_ACCEL_BVH, "tree")):
    ctx.upload(scene, accel)
    for mode, mn in ((api.RTX_MODE_DETERMINISTIC, "det"), (api.RTX_MODE_FAST, "fast")):
        out[(mn, an)] = ctx.hit(o, d, mode=mode)
ref = out[("det", "scan")]
for k, v in out.items():
    diff = (v["hit"] != ref["hit"]) | (v["prim"] != ref["prim"])
    dt = np.abs(v["t"] - ref["t"])[~diff & (ref["hit"] == 1)]
    print(name, k, "winner differs from det scan:", int(diff.sum()), "of", n, " max |dt| where equal:", float(dt.max()) if dt.size else 0.0, flush=True)
    if diff.any() and diff.sum() < 2000:
        idx = np.nonzero(diff)[0][:6]
        for i in idx:
            print("   ray", i, "det:", ref["hit"][i], ref["prim"][i], ref["t"][i], " this:", v["hit"][i], v["prim"][i], v["t"][i])
