"""The deterministic-mode contract of include/rtx_det.h (CPU only): the random stream
and the sine / cosine routines, compiled here with gcc exactly as the oracle does."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "rtx_det.h"
void det_draws(unsigned pixel, unsigned sample, unsigned long long seed, int n, float* out) {
    RtxRng r; rtx_rng_init(&r, pixel, sample, seed);
    for (int i = 0; i < n; ++i) out[i] = rtx_rng_nextf(&r);
}
void det_sincos(int n, const float* x, float* s, float* c) { for (int i = 0; i < n; ++i) rtx_sincosf(x[i], &s[i], &c[i]); }
float det_tan(float x) { return rtx_tanf(x); }
#include <math.h>
void det_atan2(int n, const float* y, const float* x, float* out) { for (int i = 0; i < n; ++i) out[i] = rtx_atan2f(y[i], x[i]); }
void det_asin(int n, const float* x, float* out) {
    for (int i = 0; i < n; ++i) out[i] = rtx_asinf_with_root(x[i], sqrtf((1.0f - x[i]) * (1.0f + x[i])));
}
"""


@pytest.fixture(scope="module")
def det(tmp_path_factory):
    d = tmp_path_factory.mktemp("det")
    (d / "det.c").write_text(SRC)
    so = d / "libdet.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    str(d / "det.c"), "-o", str(so), "-lm"], check=True)
    lib = ctypes.CDLL(str(so))
    lib.det_tan.restype = ctypes.c_float
    lib.det_tan.argtypes = [ctypes.c_float]
    return lib


def draws(lib, pixel, sample, seed, n):
    out = np.zeros(n, np.float32)
    lib.det_draws(pixel, sample, ctypes.c_uint64(seed), n, out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_draws_are_24_bit_fractions_in_unit_interval(det):
    x = draws(det, 3, 9, 42, 100000)
    assert x.min() >= 0.0 and x.max() < 1.0
    k = x * 2 ** 24
    assert np.array_equal(k, np.round(k))
    assert abs(x.mean() - 0.5) < 0.005
    hist = np.histogram(x, bins=16, range=(0, 1))[0]
    assert hist.min() > 0.9 * 100000 / 16


def test_streams_depend_on_pixel_sample_and_seed(det):
    base = draws(det, 10, 20, 30, 64)
    assert np.array_equal(base, draws(det, 10, 20, 30, 64))
    for other in (draws(det, 11, 20, 30, 64), draws(det, 10, 21, 30, 64), draws(det, 10, 20, 31, 64),
                  draws(det, 10, 20, 30 + (1 << 32), 64)):
        assert np.count_nonzero(other == base) < 4


def test_neighbouring_streams_are_uncorrelated(det):
    a = np.stack([draws(det, p, 0, 0, 8) for p in range(4096)])
    c = np.corrcoef(a[:-1, 0], a[1:, 0])[0, 1]
    assert abs(c) < 0.05
    assert abs(np.corrcoef(a[:, 0], a[:, 1])[0, 1]) < 0.05


def test_sincos_accuracy(det):
    x = np.concatenate([np.linspace(0, 2 * np.pi, 20001), np.linspace(-3000, 3000, 20001)]).astype(np.float32)
    s = np.zeros_like(x)
    c = np.zeros_like(x)
    det.det_sincos(len(x), x.ctypes.data_as(ctypes.c_void_p), s.ctypes.data_as(ctypes.c_void_p),
                   c.ctypes.data_as(ctypes.c_void_p))
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 3e-7
    assert abs(det.det_tan(0.3490658) - np.tan(0.3490658)) < 1e-6


def test_atan2_asin_accuracy(det):
    """The sphere texture coordinates' atan2 / asin (Util.h:79-80): a few ulp of the libm values
    over the whole domain, exact at the axes."""
    rng = np.random.default_rng(1)
    y = rng.normal(size=200000).astype(np.float32)
    x = rng.normal(size=200000).astype(np.float32)
    y[:8] = [0, 0, 1, -1, 1, -1, 0.0, 1e-30]
    x[:8] = [1, -1, 0, 0, 1, 1, 0.0, -1]
    out = np.zeros_like(x)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    det.det_atan2(x.size, p(y), p(x), p(out))
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(out - want).max() < 6e-7
    assert out[0] == 0.0 and out[6] == 0.0 and abs(out[2] - np.pi / 2) < 2e-7
    s = rng.uniform(-1, 1, 200000).astype(np.float32)
    s[:3] = [-1, 0, 1]
    det.det_asin(s.size, p(s), p(out))
    assert np.abs(out - np.arcsin(s.astype(np.float64))).max() < 6e-7
