"""Shared helpers for the test-suite."""
import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

REL_TOL = 1e-4  # deterministic mode: fp32 re-association tolerance on radiance


def golden_index():
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def sample_cases():
    """[(key, meta)] of the per-sample radiance fixtures made from the reference tracer."""
    return sorted(golden_index().items())


def assert_samples_match(got_rgb, got_seg, want_rgb, want_seg, exact=False):
    """Segment counts must be identical (same path geometry); radiance either bit-equal
    (`exact`, for CPU code that keeps the reference's operation order) or within REL_TOL."""
    want_seg = want_seg.astype(np.uint32)
    assert got_seg.shape == want_seg.shape
    bad = np.count_nonzero(got_seg != want_seg)
    assert bad == 0, f"{bad} of {want_seg.size} samples took a different path"
    assert np.array_equal(np.isnan(got_rgb), np.isnan(want_rgb))
    if exact:
        assert np.array_equal(got_rgb, want_rgb, equal_nan=True)
        return 0.0
    scale = np.maximum(np.abs(want_rgb), 1e-30)
    rel = np.where((got_rgb == want_rgb) | np.isnan(want_rgb), 0.0, np.abs(got_rgb - want_rgb) / scale)
    assert rel.max() <= REL_TOL, f"max relative radiance error {rel.max()}"
    return float(rel.max())


_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TREE_LIB = None


def product_tree(host_scene):
    """The scene tree of the renderer's builder as arrays (tests/tree_export.cpp, compiled here with g++
    against raytrace_amd/csrc/rtx_bvh.cpp: host code only, no GPU)."""
    global _TREE_LIB
    import ctypes as C

    from oracle import restate as _restate
    import subprocess
    import tempfile

    if _TREE_LIB is None:
        so = os.path.join(tempfile.mkdtemp(prefix="tree_export_"), "libtree_export.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(_ROOT, "include"),
                        "-I" + os.path.join(_ROOT, "raytrace_amd", "csrc"), os.path.join(_ROOT, "tests", "tree_export.cpp"),
                        os.path.join(_ROOT, "raytrace_amd", "csrc", "rtx_bvh.cpp"), "-o", so], check=True)
        _TREE_LIB = C.CDLL(so)
        _TREE_LIB.tree_export_build.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                                C.POINTER(C.c_double)]
        _TREE_LIB.tree_export_copy.argtypes = [C.c_void_p] * 8
    n_items, n_big, depth, visits = C.c_uint32(0), C.c_uint32(0), C.c_int32(0), C.c_double(0)
    n = _TREE_LIB.tree_export_build(_restate._scene_ptr(host_scene), C.byref(n_items), C.byref(n_big), C.byref(depth), C.byref(visits))
    assert n > 0
    t = dict(lo=np.zeros((n, 3), np.float32), hi=np.zeros((n, 3), np.float32), left=np.zeros(n, np.int32),
             right=np.zeros(n, np.int32), first=np.zeros(n, np.int32), count=np.zeros(n, np.int32),
             item_begin=np.zeros(n_items.value, np.uint32), item_end=np.zeros(n_items.value, np.uint32))
    _TREE_LIB.tree_export_copy(*[t[k].ctypes.data for k in ("lo", "hi", "left", "right", "first", "count", "item_begin", "item_end")])
    t.update(n_big=n_big.value, depth=depth.value, visits=visits.value)
    return t


