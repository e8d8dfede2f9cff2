"""Surface-area estimates of both trees of named scenes (GPU box: the upload needs a device).
usage: RTX_TREE_STATS=1 python tools/tree_stats.py scene ...   (one line per scene on stderr)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytrace_amd import api
ctx = api.Context(0)
for name in sys.argv[1:]:
    sys.stderr.write(name + ": "); sys.stderr.flush()
    ctx.upload(api.HostScene(name, 64, 36), api.RTX_ACCEL_BVH)
