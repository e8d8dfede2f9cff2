"""Depth and surface-area estimate of the scene tree of named scenes (GPU box: the upload needs a device).
usage: python tools/tree_stats.py scene ...   (one line per scene)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from raytrace_amd import api
ctx = api.Context(0)
for name in sys.argv[1:]:
    ctx.upload(api.HostScene(name, 64, 36), api.RTX_ACCEL_BVH)
    i = ctx.info()
    print(f"{name}: {i.n_items} items, {i.n_bvh_nodes} nodes, depth {i.bvh_depth}, {i.bvh_visits:.2f} estimated node visits per ray")
