"""End to end through the host scene API on the GPU: Scene::render() of the drop-in classes
(build -> flatten -> rtx_* -> filter chain -> BMP file; the counterpart of Scene.cpp:134-156 driven
by main.cpp:8-12) against the reference's own pipeline on the same stream: Scene::color per sample,
c /= ns, Image::write, stbi_write_bmp."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from raytrace_amd import api

pytestmark = pytest.mark.gpu


def _reference_bmp(tmp_path, name, w, h, spp, seed):
    from oracle import ref

    if not ref.available():
        pytest.skip("oracle/_ref/libref_oracle.so not built")
    want = ref.RefScene(name, w, h).render(0, spp, seed=seed, threads=8)
    path = str(tmp_path / "reference.bmp")
    assert ref.write_bmp(path, ref.resolve(want["sum"], spp))
    return open(path, "rb").read(), want["rays"]


def _check_bmp(got, want, w, h):
    assert len(got) == len(want)
    header = 54
    assert got[:header] == want[:header]  # BITMAPFILEHEADER + BITMAPINFOHEADER as stbi_write_bmp lays them out
    row = (3 * w + 3) // 4 * 4
    g = np.frombuffer(got, np.uint8)[header:].reshape(h, row)
    r = np.frombuffer(want, np.uint8)[header:].reshape(h, row)
    assert np.array_equal(g[:, 3 * w:], r[:, 3 * w:])  # row padding
    # pixel bytes: the radiance sum is associated differently on the GPU (iterative weight product),
    # which moves a byte by one code where the mean sits on a rounding edge
    d = np.abs(g[:, :3 * w].astype(int) - r[:, :3 * w].astype(int))
    assert d.max() <= 1
    assert np.count_nonzero(d) <= 0.001 * d.size


@pytest.mark.parametrize("name,w,h,spp", [("cornell_builtin", 64, 36, 8), ("cornell_extra", 50, 30, 6)])
def test_scene_render_writes_the_references_bmp(gpu_ctx, tmp_path, name, w, h, spp):
    lib = api.load_libhost()
    want, want_rays = _reference_bmp(tmp_path, name, w, h, spp, seed=5)
    out = str(tmp_path / "gpu.bmp")
    rays, seconds = C.c_ulonglong(0), C.c_double(0.0)
    ok = lib.rtxh_render_to_bmp(name.encode(), out.encode(), w, h, spp, 1, 5, 0, C.byref(rays), C.byref(seconds))
    assert ok, lib.rtxh_last_error().decode()
    assert rays.value == want_rays
    _check_bmp(open(out, "rb").read(), want, w, h)


def test_scene_render_over_two_contexts_writes_the_same_file(gpu_ctx, tmp_path):
    """Scene::setDevices / `rtx_render --devices 0,0`: the frame split by sample range over two
    contexts (here on the one GPU of the box) and summed by rtx_render_multi."""
    lib = api.load_libhost()
    lib.rtxh_render_to_bmp_multi.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                             C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_double)]
    lib.rtxh_render_to_bmp_multi.restype = C.c_int
    name, w, h, spp = "cornell_builtin", 64, 36, 9
    want, want_rays = _reference_bmp(tmp_path, name, w, h, spp, seed=6)
    out = str(tmp_path / "gpu2.bmp")
    rays, seconds = C.c_ulonglong(0), C.c_double(0.0)
    devices = (C.c_int * 2)(0, 0)
    ok = lib.rtxh_render_to_bmp_multi(name.encode(), out.encode(), w, h, spp, 1, 6, devices, 2, C.byref(rays), C.byref(seconds))
    assert ok, lib.rtxh_last_error().decode()
    assert rays.value == want_rays
    _check_bmp(open(out, "rb").read(), want, w, h)
    # fast mode (the default of main.cpp's counterpart): a plausible image of the same scene
    # (against the deterministic mode at the same sample count: gamma and clamp make the mean of
    # the bytes depend on the noise level)
    fast, det = str(tmp_path / "fast.bmp"), str(tmp_path / "det.bmp")
    assert lib.rtxh_render_to_bmp(name.encode(), fast.encode(), w, h, 256, 0, 6, 0, C.byref(rays), C.byref(seconds))
    assert lib.rtxh_render_to_bmp(name.encode(), det.encode(), w, h, 256, 1, 6, 0, C.byref(rays), C.byref(seconds))
    f = np.frombuffer(open(fast, "rb").read(), np.uint8)[54:].astype(float).mean()
    r = np.frombuffer(open(det, "rb").read(), np.uint8)[54:].astype(float).mean()
    assert abs(f / r - 1) < 0.03


def test_scene_render_with_the_kernels_compiled_for_the_scene(gpu_ctx, tmp_path):
    """Scene::setKernelCompilation / `rtx_render --compile`: the same fast-mode image, up to the noise of a few
    samples whose paths differ in rounding, from the table-driven kernels and from the compiled ones; the
    deterministic mode is untouched by the setting."""
    lib = api.load_libhost()
    name, w, h, spp = "cornell_builtin", 64, 36, 64
    rays, seconds = C.c_ulonglong(0), C.c_double(0.0)
    files, counts = {}, {}
    try:
        for label, mode in (("never", 0), ("upload", 1), ("background", 2), ("auto", -1)):
            assert lib.rtxh_set_kernel_compilation(mode)
            files[label] = str(tmp_path / f"{label}.bmp")
            assert lib.rtxh_render_to_bmp(name.encode(), files[label].encode(), w, h, spp, 0, 4, 0, C.byref(rays), C.byref(seconds)), \
                lib.rtxh_last_error().decode()
            counts[label] = rays.value
        det = {}
        for label, mode in (("never", 0), ("upload", 1)):
            assert lib.rtxh_set_kernel_compilation(mode)
            det[label] = str(tmp_path / f"det_{label}.bmp")
            assert lib.rtxh_render_to_bmp(name.encode(), det[label].encode(), w, h, 8, 1, 4, 0, C.byref(rays), C.byref(seconds))
        assert not lib.rtxh_set_kernel_compilation(5)
    finally:
        lib.rtxh_set_kernel_compilation(-1)
    image = {k: np.frombuffer(open(v, "rb").read(), np.uint8)[54:].astype(int) for k, v in files.items()}
    # a small frame is below the size from which "auto" compiles: the table-driven kernels, bit for bit
    assert np.array_equal(image["auto"], image["never"]) and counts["auto"] == counts["never"]
    for label in ("upload", "background"):
        d = np.abs(image[label] - image["never"])
        assert (d > 0).mean() < 0.05 and abs(image[label].mean() / image["never"].mean() - 1) < 5e-3, label
        assert abs(counts[label] / counts["never"] - 1) < 2e-3
    assert open(det["never"], "rb").read() == open(det["upload"], "rb").read()


def test_a_process_may_exit_while_its_kernels_still_compile(gpu_ctx, tmp_path):
    """Background compilation (`rtx_render --compile background`, RTX_OPTION_SPECIALIZE_SCAN = 2) of a frame that
    is finished long before the compiler: the context waits for its compilation when it is destroyed, and a
    context that is never destroyed is waited for when the process exits - a compiler thread still running while
    the process's libraries are torn down ends in heap corruption (exit status 134)."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "raytrace_amd", "rtx_render")
    if not os.path.exists(tool):
        pytest.skip("raytrace_amd/rtx_render not built")
    out = subprocess.run([tool, "--scene", "cornell_classic", "--width", "48", "--height", "27", "--spp", "2", "--out",
                          str(tmp_path / "tiny.bmp"), "--compile", "background"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-400:]
    leak = ("import sys; sys.path.insert(0, %r)\n"
            "from raytrace_amd import api\n"
            "ctx = api.Context(0)\n"
            "ctx.set_option(api.RTX_OPTION_SPECIALIZE_SCAN, 2)\n"
            "ctx.upload(api.HostScene('small', 48, 27), api.RTX_ACCEL_LINEAR)\n"
            "print(ctx.note(), flush=True)\n"
            "ctx._ctx = None\n" % root)  # (no rtx_destroy)
    out = subprocess.run([sys.executable, "-c", leak], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-400:]
    assert "compiling in the background" in out.stdout or out.stdout.strip() == ""
