"""Event counts of the fast-mode BVH trace kernel (GPU box; needs a library built with
-DRTX_BVH_STATS: make -C raytrace_amd/csrc EXTRA=-DRTX_BVH_STATS OUT=../librtx_stats.so BUILD=build_stats,
then RTX_LIBRTX=raytrace_amd/librtx_stats.so python tools/bvh_stats.py scene w h spp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api

name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = api.Context(0)
ctx.upload(api.HostScene(name, w, h), api.RTX_ACCEL_BVH)
acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
c = torch.zeros(13, dtype=torch.int64, device="cuda:0")
ctx.render_device(ctx.params(w, h, 0, spp, seed=1), acc.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
rays, wn, ln, wi, li, ws, wsh, lsh, ldone, lidle, lleaf, lempty_hit, lempty = [int(x) for x in c.tolist()]
print(f"{name} {w}x{h}x{spp}: rays {rays}")
print(f"node steps: {ln / rays:.2f} per ray; wave steps {wn}, lanes per step {ln / max(wn, 1):.1f}")
print(f"item tests: {li / rays:.2f} per ray; wave steps {wi}, lanes per step {li / max(wi, 1):.1f}")
print(f"walk-loop iterations {ws} ({ws * 64 / rays:.2f} lane slots per ray); shade passes {wsh}, lanes per pass {lsh / max(wsh, 1):.1f}")
print(f"lanes per walk-loop iteration: at a node {ln / ws:.1f}, holding items {lleaf / ws:.1f}, waiting to be shaded {ldone / ws:.1f}, "
      f"without an item {lidle / ws:.1f}")
print(f"node steps that met no child: {lempty / max(ln, 1):.3f} of all ({lempty / rays:.2f} per ray); of these with a hit already in hand: "
      f"{lempty_hit / max(lempty, 1):.3f}")
