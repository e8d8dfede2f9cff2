"""ctypes wrapper of oracle/librestate.so, the CPU restatement of the reference
algorithm over a flattened scene (oracle/restate.cpp).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librestate.so")
_lib = None

HIT_DTYPE = np.dtype([("hit", "<i4"), ("prim", "<i4"), ("t", "<f4"),
                      ("p", "<f4", (3,)), ("n", "<f4", (3,)), ("material", "<i4"), ("uv", "<f4", (2,))])


def available() -> bool:
    return os.path.exists(LIB_PATH)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        lib = C.CDLL(LIB_PATH)
        lib.restate_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        lib.restate_render.restype = C.c_int
        lib.restate_count_flops.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                            C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.restate_count_flops.restype = C.c_int
        lib.restate_hit.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        lib.restate_hit.restype = C.c_int
        lib.restate_image_write.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        lib.restate_set_bvh.argtypes = [C.c_int]
        lib.restate_set_bvh.restype = None
        _lib = lib
    return _lib


def _scene_ptr(scene):
    """Accepts a raytrace_amd.api.HostScene or a ctypes RtxScene."""
    if hasattr(scene, "view_ptr"):
        return C.cast(scene.view_ptr, C.c_void_p)
    return C.cast(C.pointer(scene), C.c_void_p)


class _accel:
    """Scope in which the restatement walks its own BVH instead of scanning the list."""

    def __init__(self, bvh):
        self.bvh = bool(bvh)

    def __enter__(self):
        load().restate_set_bvh(1 if self.bvh else 0)

    def __exit__(self, *exc):
        load().restate_set_bvh(0)


def render(scene, w, h, spp_begin, spp_count, seed=0, threads=1, per_sample=False, max_depth=50, bvh=False, tree=None):
    with _accel(bvh), _external_tree(tree):
        return _render(scene, w, h, spp_begin, spp_count, seed, threads, per_sample, max_depth)


def _render(scene, w, h, spp_begin, spp_count, seed, threads, per_sample, max_depth):
    lib = load()
    n = w * h
    acc = np.zeros((h, w, 3), np.float32)
    rays = C.c_uint64(0)
    samples = np.zeros((n, spp_count, 3), np.float32) if per_sample else None
    segs = np.zeros((n, spp_count), np.uint32) if per_sample else None
    rc = lib.restate_render(_scene_ptr(scene), w, h, spp_begin, spp_count, seed, max_depth, threads, acc.ctypes.data,
                            samples.ctypes.data if per_sample else None, segs.ctypes.data if per_sample else None,
                            C.byref(rays))
    assert rc == 0
    out = {"sum": acc, "rays": rays.value}
    if per_sample:
        out["samples"], out["segments"] = samples, segs
    return out


class _external_tree:
    """Scope in which the nearest-hit query walks a tree handed in from outside (a dict of arrays as
    tests/tree_export.cpp hands them out: item_begin, item_end, n_big, lo, hi, left, right, first,
    count) the way the GPU kernel walks its two-child nodes.  None: no change."""

    def __init__(self, tree):
        self.tree = tree

    def __enter__(self):
        if self.tree is None:
            return
        lib = load()
        lib.restate_set_external_tree.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
        lib.restate_set_external_tree.restype = None
        lib.restate_last_tree_nodes.restype = C.c_ulonglong
        t = self.tree
        a = {k: np.ascontiguousarray(t[k]) for k in ("item_begin", "item_end", "lo", "hi", "left", "right", "first", "count")}
        lib.restate_set_external_tree(len(a["item_begin"]), a["item_begin"].ctypes.data, a["item_end"].ctypes.data,
                                      int(t["n_big"]), len(a["left"]), a["lo"].ctypes.data, a["hi"].ctypes.data,
                                      a["left"].ctypes.data, a["right"].ctypes.data, a["first"].ctypes.data,
                                      a["count"].ctypes.data)

    def __exit__(self, *exc):
        if self.tree is not None:
            load().restate_set_external_tree(0, None, None, 0, 0, None, None, None, None, None, None)


def count_flops(scene, w, h, spp_begin, spp_count, seed=0, threads=1, max_depth=50, bvh=False, tree=None):
    """Counted fp32 operations and rays of a render; with `tree` (see _external_tree) also the inner
    nodes the queries visited, as "tree_nodes"."""
    lib = load()
    rays, flops = C.c_uint64(0), C.c_uint64(0)
    with _accel(bvh), _external_tree(tree):
        rc = lib.restate_count_flops(_scene_ptr(scene), w, h, spp_begin, spp_count, seed, max_depth, threads,
                                     C.byref(rays), C.byref(flops))
        assert rc == 0
        out = {"rays": rays.value, "flops": flops.value}
        if tree is not None:
            out["tree_nodes"] = int(lib.restate_last_tree_nodes())
    return out


def hit(scene, origins, directions, t0=0.001, t1=3.402823466e+38, bvh=False):
    lib = load()
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
    hits = np.zeros(o.shape[0], HIT_DTYPE)
    with _accel(bvh):
        rc = lib.restate_hit(_scene_ptr(scene), o.shape[0], o.ctypes.data, d.ctypes.data, t0, t1, hits.ctypes.data)
    assert rc == 0
    return hits


def image_write(rgb):
    lib = load()
    a = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), np.uint8)
    lib.restate_image_write(a.shape[0], a.ctypes.data, out.ctypes.data)
    return out


def background_sky(dirs):
    """Scene::backgroundSky (Scene.h:24-28) of n directions."""
    lib = load()
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    out = np.zeros_like(d)
    lib.restate_background_sky(C.c_int(d.shape[0]), C.c_void_p(d.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def light_pdf(scene, origins, directions, seed=0):
    """ShapePdf::value / generate of the scene's light shape for n (origin, direction) pairs; entry
    k draws from the stream of (pixel k, sample 0, seed).  Returns (values, directions)."""
    lib = load()
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
    vals = np.zeros(o.shape[0], np.float32)
    dirs = np.zeros_like(o)
    rc = lib.restate_light_pdf(_scene_ptr(scene), C.c_int(o.shape[0]), C.c_void_p(o.ctypes.data),
                               C.c_void_p(d.ctypes.data), C.c_uint64(seed), C.c_void_p(vals.ctypes.data),
                               C.c_void_p(dirs.ctypes.data))
    assert rc == 0
    return vals, dirs


def cosine_pdf(normals, seed=0):
    """CosinePdf::generate / value (CosinePdf.cpp:6-20) around n normals; entry k draws from the
    stream of (pixel k, sample 0, seed).  Returns (directions, values)."""
    lib = load()
    nn = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    dirs = np.zeros_like(nn)
    vals = np.zeros(nn.shape[0], np.float32)
    lib.restate_cosine_pdf(C.c_int(nn.shape[0]), C.c_void_p(nn.ctypes.data), C.c_uint64(seed),
                            C.c_void_p(dirs.ctypes.data), C.c_void_p(vals.ctypes.data))
    return dirs, vals


def scatter(kind, param, tex, d_in, normals, scattered, seed=0):
    """Material::scatter / emitted / scattering_pdf of one material (kind: 0 Lambertian, 1 Metal,
    2 Dielectric, 3 DiffuseLight; param = fuzz | refractive index; tex = its plain colour) for n
    (incoming direction, hit normal, scattered direction) triples; entry k draws from the stream
    of (pixel k, sample 0, seed).  Returns dict(ok, specular, dir, albedo, emitted, spdf)."""
    lib = load()
    d = np.ascontiguousarray(d_in, np.float32).reshape(-1, 3)
    nn = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    sd = np.ascontiguousarray(scattered, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tex, np.float32).reshape(3)
    n = d.shape[0]
    out = {"ok": np.zeros(n, np.int32), "specular": np.zeros(n, np.int32), "dir": np.zeros((n, 3), np.float32),
           "albedo": np.zeros((n, 3), np.float32), "emitted": np.zeros((n, 3), np.float32), "spdf": np.zeros(n, np.float32)}
    rc = lib.restate_scatter(C.c_uint32(kind), C.c_float(param), C.c_void_p(t.ctypes.data), C.c_int(n), C.c_void_p(d.ctypes.data),
                        C.c_void_p(nn.ctypes.data), C.c_void_p(sd.ctypes.data), C.c_uint64(seed),
                        *[C.c_void_p(out[k].ctypes.data) for k in ("ok", "specular", "dir", "albedo", "emitted", "spdf")])
    assert rc == 0
    return out
