"""Host-side cost of a scene: building + flattening it (librtx_host) and rtx_upload_scene (repacking,
BVH build, light tree, copies to HBM), which the throughput metric excludes.  GPU box."""
import time, sys
sys.path.insert(0, '.')
from raytrace_amd import api
for name in ("cornell", "random_field", "bvh_stress", "cornell_many_lights_40"):
    t = time.time(); s = api.HostScene(name, 1920, 1080); t1 = time.time()
    c = api.Context(0); c.upload(s, api.RTX_ACCEL_AUTO); t2 = time.time()
    c.upload(s, api.RTX_ACCEL_AUTO); t3 = time.time()
    print(name, "host build+flatten %.3f s, first upload %.3f s, second upload %.3f s" % (t1 - t, t2 - t1, t3 - t2), c.info().n_items)
