"""Ad-hoc GPU probe: parity statistics and first throughput numbers."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import api
from oracle import ref

ctx = api.Context(0)
for name, w, h, spp, accel in [("cornell", 64, 36, 16, 1), ("random_field", 48, 27, 4, 2), ("cornell_extra", 64, 36, 8, 2)]:
    sc = api.HostScene(name, w, h)
    ctx.upload(sc, accel)
    p = ctx.params(w, h, 0, spp, mode=1, seed=7)
    rgb, seg = ctx.render_samples(p)
    want = ref.RefScene(name, w, h).render(0, spp, seed=7, threads=8, per_sample=True)
    scale = np.maximum(np.abs(want["samples"]), 1e-30)
    rel = np.abs(rgb - want["samples"]) / scale
    nz = want["samples"] > 0
    print(name, "seg equal", np.array_equal(seg, want["segments"]), "max seg", seg.max(), "mean seg", seg.mean(),
          "nonzero frac", nz.mean(), "max rel", rel[nz].max() if nz.any() else 0, "bit-equal frac", (rgb == want["samples"]).mean())
    pf = ctx.params(w, h, 0, spp, mode=0, seed=7)
    rgbf, segf = ctx.render_samples(pf)
    print("   fast: seg equal frac", (segf == seg).mean(), "mean radiance det", rgb.mean(), "fast", rgbf.mean())

for name, accel, spp in [("cornell", 1, 64), ("cornell", 2, 16), ("random_field", 2, 16), ("bvh_stress", 2, 4)]:
    w, h = 1920, 1080
    t = time.time(); sc = api.HostScene(name, w, h); tb = time.time() - t
    t = time.time(); ctx.upload(sc, accel); tu = time.time() - t
    info = ctx.info()
    for mode in (0, 1):
        p = ctx.params(w, h, 0, spp, mode=mode, seed=1)
        ctx.render(ctx.params(w, h, 0, 1, mode=mode, seed=1))
        t = time.time(); acc, rays = ctx.render(p); dt = time.time() - t
        print(f"{name} accel={info.accel} nodes={info.n_bvh_nodes} build {tb:.2f}s upload {tu:.2f}s mode={mode} spp={spp}: {dt*1e3:.1f} ms "
              f"rays {rays} ({rays/(w*h*spp):.3f}/sample) {rays/dt/1e9:.3f} Grays/s mean {acc.mean()/spp:.4f} nan {np.isnan(acc).sum()}", flush=True)
