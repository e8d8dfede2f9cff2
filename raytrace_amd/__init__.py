"""MI355X-native path tracer: HIP kernels behind a C ABI plus a host scene API.

``raytrace_amd.api`` binds the C ABI of ``include/rtx.h`` (``librtx.so``) and the
host scene factory (``librtx_host.so``) with ctypes.
"""
from . import api  # noqa: F401

__all__ = ["api"]
