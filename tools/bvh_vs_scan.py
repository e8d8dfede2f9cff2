import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from raytrace_amd import api
ctx = api.Context(0)
for name in sys.argv[1:]:
    w, h, spp = 48, 27, 4
    scene = api.HostScene(name, w, h)
    p = ctx.params(w, h, 3, spp, mode=api.RTX_MODE_FAST, seed=31)
    ctx.upload(scene, api.RTX_ACCEL_LINEAR)
    lr, ls = ctx.render_samples(p)
    ctx.upload(scene, api.RTX_ACCEL_BVH)
    br, bs = ctx.render_samples(p)
    same = bs == ls
    close = np.isclose(br, lr, rtol=2e-2, atol=1e-3).all(axis=-1)
    print(name, "nodes", ctx.info().n_bvh_nodes, "path mismatches", int((~same).sum()), "of", same.size, "radiance mismatches on same paths", int((~close & same).sum()),
          "mean segs", float(ls.mean()), flush=True)
