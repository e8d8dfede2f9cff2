"""rtx_render_noalloc: rtx_render over the library's file-scope device array instead of buffers
the context allocates.  Same kernels, same order of additions: its output must be the output of
rtx_render bit for bit, on two calls in a row (the second smaller, so that what the first call
left in the shared array would show), in both arithmetic modes, with and without `accumulate`.
"""
import numpy as np
import pytest

from raytrace_amd import api

pytestmark = pytest.mark.gpu


def test_noalloc_render_equals_rtx_render_on_two_calls_in_a_row(gpu_ctx):
    for mode in (api.RTX_MODE_DETERMINISTIC, api.RTX_MODE_FAST):
        for name, accel in (("cornell", api.RTX_ACCEL_LINEAR), ("random_field", api.RTX_ACCEL_BVH)):
            first = None
            for w, h, spp in ((96, 54, 6), (40, 23, 3)):  # the second call is the smaller one
                scene = api.HostScene(name, w, h)
                gpu_ctx.upload(scene, accel)
                p = gpu_ctx.params(w, h, 2, spp, mode=mode, seed=21)
                got, got_rays = gpu_ctx.render_noalloc(p)
                want, want_rays = gpu_ctx.render(p)
                assert got_rays == want_rays and got_rays >= w * h * spp
                assert np.array_equal(got, want)
                if first is None:
                    first = got
            assert first.any()
            # accumulate: the caller's frame is the starting value, whatever the array held before
            w, h = 40, 23
            start = np.full((h, w, 3), 0.25, np.float32)
            p = gpu_ctx.params(w, h, 5, 2, mode=mode, seed=21, accumulate=True)
            got, got_rays = gpu_ctx.render_noalloc(p, start.copy())
            want, want_rays = gpu_ctx.render(p, start.copy())
            assert got_rays == want_rays
            assert np.array_equal(got, want)


def test_noalloc_render_refuses_frames_beyond_its_static_buffer(gpu_ctx):
    scene = api.HostScene("cornell", 16, 9)
    gpu_ctx.upload(scene, api.RTX_ACCEL_LINEAR)
    w, h = 2048, 1025  # one row more than 2^21 pixels
    assert w * h > api.RTX_NOALLOC_MAX_PIXELS
    with pytest.raises(api.RtxError) as e:
        gpu_ctx.render_noalloc(gpu_ctx.params(w, h, 0, 1))
    assert e.value.code == api.RTX_ERR_TOO_LARGE
    assert "static device buffer" in str(e.value)
    # the largest frame it takes: exactly 2^21 pixels (1920 x 1080 lies below that)
    big, rays = gpu_ctx.render_noalloc(gpu_ctx.params(2048, 1024, 0, 1))
    assert rays >= 2048 * 1024 and np.isfinite(big).all()
    assert np.array_equal(big, gpu_ctx.render(gpu_ctx.params(2048, 1024, 0, 1))[0])
