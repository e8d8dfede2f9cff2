import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the image file of the "textured" scene (raytrace_amd/host/scenes.inc reads it by this name)
os.environ.setdefault("RTX_TEXTURE", os.path.join(ROOT, "tests", "golden", "texture_64x32.bmp"))


def _jpeg_texture():
    """The JPEG file of the "textured_jpeg" scene: the reference's texture asset, kept as bytes in
    the fixture tests/golden/jpeg_cases.npz beside the texels the reference's loader makes of it."""
    import atexit
    import tempfile

    import numpy as np

    fixture = os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz")
    if "RTX_TEXTURE_JPEG" in os.environ or not os.path.exists(fixture):
        return
    fd, path = tempfile.mkstemp(suffix=".jpg", prefix="rtx_texture_")
    with os.fdopen(fd, "wb") as f:
        f.write(np.load(fixture)["asset_field001_file"].tobytes())
    atexit.register(lambda: os.path.exists(path) and os.remove(path))
    os.environ["RTX_TEXTURE_JPEG"] = path


_jpeg_texture()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One RtxContext on device 0 for the whole GPU test session."""
    from raytrace_amd import api

    if api.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    ctx = api.Context(0)
    yield ctx
    ctx.close()
