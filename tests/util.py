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
