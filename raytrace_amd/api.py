"""ctypes bindings of the C ABI (include/rtx.h) and of the host scene factory.

The bindings are deliberately thin: every call goes straight through the
``extern "C"`` entry points of ``librtx.so`` (HIP kernels) and
``librtx_host.so`` (host scene API).  There is no Python or CPU implementation of
the renderer here; if the shared libraries are missing, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))

RTX_ABI_VERSION = 5
RTX_ACCEL_AUTO, RTX_ACCEL_LINEAR, RTX_ACCEL_BVH = 0, 1, 2
RTX_STEP_TRANSLATE, RTX_STEP_ROTATE = 0, 1
RTX_MODE_FAST, RTX_MODE_DETERMINISTIC = 0, 1
RTX_PRIM_RECT_XY, RTX_PRIM_RECT_XZ, RTX_PRIM_RECT_YZ, RTX_PRIM_SPHERE = 0, 1, 2, 3
RTX_MAT_LAMBERTIAN, RTX_MAT_METAL, RTX_MAT_DIELECTRIC, RTX_MAT_DIFFUSE_LIGHT = 0, 1, 2, 3
RTX_TEX_COLOR, RTX_TEX_CHECKER, RTX_TEX_IMAGE, RTX_TEX_SKY = 0, 1, 2, 3
RTX_BACKGROUND_CONSTANT, RTX_BACKGROUND_SKY = 0, 1
RTX_LIGHT_RECT_XY, RTX_LIGHT_RECT_XZ, RTX_LIGHT_RECT_YZ, RTX_LIGHT_SPHERE, RTX_LIGHT_OTHER, RTX_LIGHT_LIST = range(6)
RTX_LIGHTS_REFERENCE, RTX_LIGHTS_GENERAL = 0, 1
RTX_OK = 0
RTX_ERR_INVALID_ARGUMENT, RTX_ERR_NO_DEVICE, RTX_ERR_HIP = -1, -2, -3
RTX_ERR_NO_SCENE, RTX_ERR_UNSUPPORTED, RTX_ERR_TOO_LARGE = -4, -5, -6

_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)


class RtxScene(C.Structure):
    """Mirror of ``struct RtxScene`` (include/rtx.h)."""

    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_prims", C.c_uint32),
        ("prim_kind", _u32p), ("prim_flip", _u32p), ("prim_material", _u32p), ("prim_chain", _u32p),
        ("prim_a", _f32p), ("prim_b", _f32p), ("prim_c", _f32p), ("prim_d", _f32p), ("prim_e", _f32p),
        ("n_chains", C.c_uint32),
        ("chain_begin", _u32p),
        ("n_steps", C.c_uint32),
        ("step_kind", _u32p),
        ("step_x", _f32p), ("step_y", _f32p), ("step_z", _f32p), ("step_w", _f32p),
        ("n_materials", C.c_uint32),
        ("mat_kind", _u32p), ("mat_texture", _u32p), ("mat_param", _f32p),
        ("n_textures", C.c_uint32),
        ("tex_kind", _u32p),
        ("tex_r", _f32p), ("tex_g", _f32p), ("tex_b", _f32p),
        ("tex_odd", _u32p), ("tex_even", _u32p), ("tex_offset", _u32p),
        ("n_texel_bytes", C.c_uint64),
        ("texels", C.POINTER(C.c_uint8)),
        ("n_lights", C.c_uint32),
        ("light_is_list", C.c_uint32),
        ("light_kind", _u32p),
        ("light_a", _f32p), ("light_b", _f32p), ("light_c", _f32p), ("light_d", _f32p), ("light_e", _f32p),
        ("cam_origin", C.c_float * 3), ("cam_ll", C.c_float * 3),
        ("cam_u", C.c_float * 3), ("cam_v", C.c_float * 3),
        ("background", C.c_float * 3),
        ("background_mode", C.c_uint32),
        ("n_light_entries", C.c_uint32),
        ("light_sampling", C.c_uint32),
        ("light_first", _u32p), ("light_count", _u32p), ("light_chain", _u32p),
    ]


class RtxParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("spp_begin", C.c_uint32), ("spp_count", C.c_uint32),
        ("max_depth", C.c_uint32), ("mode", C.c_uint32),
        ("seed", C.c_uint64),
        ("accumulate", C.c_uint32), ("reserved", C.c_uint32),
    ]


class RtxHit(C.Structure):
    _fields_ = [
        ("hit", C.c_int32), ("prim", C.c_int32), ("t", C.c_float),
        ("p", C.c_float * 3), ("n", C.c_float * 3), ("material", C.c_int32),
        ("uv", C.c_float * 2),
    ]


class RtxSceneInfo(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("n_items", C.c_uint32),
                ("n_bvh_nodes", C.c_uint32), ("accel", C.c_uint32),
                ("bvh_depth", C.c_uint32), ("bvh_visits", C.c_float),
                ("scan_specialized", C.c_uint32), ("specialize_seconds", C.c_float)]


RTX_OPTION_SPECIALIZE_SCAN = 1


HIT_DTYPE = np.dtype([("hit", "<i4"), ("prim", "<i4"), ("t", "<f4"),
                      ("p", "<f4", (3,)), ("n", "<f4", (3,)), ("material", "<i4"), ("uv", "<f4", (2,))])

#: every symbol include/rtx.h declares
RTX_SYMBOLS = (
    "rtx_create", "rtx_destroy", "rtx_last_error", "rtx_device_count", "rtx_upload_scene",
    "rtx_render", "rtx_render_device", "rtx_render_samples", "rtx_hit", "rtx_resolve_device",
    "rtx_scene_info", "rtx_render_multi", "rtx_render_noalloc", "rtx_resolve_to_device",
    "rtx_set_option", "rtx_prepare_scene", "rtx_scene_note", "rtx_scene_kernel_source",
)
RTX_NOALLOC_MAX_PIXELS = 1 << 21
RTXH_SYMBOLS = (
    "rtxh_last_error", "rtxh_scene_create", "rtxh_scene_view", "rtxh_scene_destroy",
    "rtxh_image_write", "rtxh_resolve", "rtxh_write_bmp", "rtxh_render_to_bmp",
    "rtxh_render_to_bmp_multi", "rtxh_load_image", "rtxh_set_kernel_compilation",
)


class RtxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rtx error {code}: {message}")
        self.code = code


_librtx = None
_libhost = None


def librtx_path() -> str:
    # RTX_LIBRTX selects another build of the same library (A/B timing of kernel variants)
    return os.environ.get("RTX_LIBRTX") or os.path.join(_PKG_DIR, "librtx.so")


def libhost_path() -> str:
    return os.path.join(_PKG_DIR, "librtx_host.so")


def load_librtx() -> C.CDLL:
    """Load the HIP library.  Raises OSError if it has not been built."""
    global _librtx
    if _librtx is None:
        # PyTorch wheels bundle their own HIP runtime (same SONAME as /opt/rocm's).  Two HIP
        # runtimes in one process do not both see the GPU, so when torch is installed it must be
        # loaded FIRST; librtx.so then binds to the runtime already in the process.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = librtx_path()
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: build it with `make -C raytrace_amd/csrc` "
                          "(there is no fallback implementation)")
        lib = C.CDLL(path)
        ctxp = C.c_void_p
        lib.rtx_create.argtypes = [C.c_int, C.POINTER(ctxp)]
        lib.rtx_create.restype = C.c_int
        lib.rtx_destroy.argtypes = [ctxp]
        lib.rtx_destroy.restype = None
        lib.rtx_last_error.argtypes = []
        lib.rtx_last_error.restype = C.c_char_p
        lib.rtx_device_count.argtypes = []
        lib.rtx_device_count.restype = C.c_int
        lib.rtx_upload_scene.argtypes = [ctxp, C.POINTER(RtxScene), C.c_uint32]
        lib.rtx_upload_scene.restype = C.c_int
        lib.rtx_render.argtypes = [ctxp, C.POINTER(RtxParams), C.c_void_p, C.POINTER(C.c_uint64)]
        lib.rtx_render.restype = C.c_int
        lib.rtx_render_noalloc.argtypes = [ctxp, C.POINTER(RtxParams), C.c_void_p, C.POINTER(C.c_uint64)]
        lib.rtx_render_noalloc.restype = C.c_int
        lib.rtx_render_multi.argtypes = [C.POINTER(ctxp), C.c_uint32, C.POINTER(RtxParams), C.c_void_p,
                                         C.POINTER(C.c_uint64)]
        lib.rtx_render_multi.restype = C.c_int
        lib.rtx_render_device.argtypes = [ctxp, C.POINTER(RtxParams), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rtx_render_device.restype = C.c_int
        lib.rtx_render_samples.argtypes = [ctxp, C.POINTER(RtxParams), C.c_void_p, C.c_void_p]
        lib.rtx_render_samples.restype = C.c_int
        lib.rtx_hit.argtypes = [ctxp, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                C.c_uint32, C.c_void_p]
        lib.rtx_hit.restype = C.c_int
        lib.rtx_resolve_device.argtypes = [ctxp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        lib.rtx_resolve_device.restype = C.c_int
        lib.rtx_resolve_to_device.argtypes = [ctxp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rtx_resolve_to_device.restype = C.c_int
        lib.rtx_scene_info.argtypes = [ctxp, C.POINTER(RtxSceneInfo)]
        lib.rtx_scene_info.restype = C.c_int
        lib.rtx_set_option.argtypes = [ctxp, C.c_uint32, C.c_uint64]
        lib.rtx_set_option.restype = C.c_int
        lib.rtx_prepare_scene.argtypes = [C.POINTER(RtxScene)]
        lib.rtx_prepare_scene.restype = C.c_int
        lib.rtx_scene_kernel_source.argtypes = [C.POINTER(RtxScene), C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.rtx_scene_kernel_source.restype = C.c_int
        lib.rtx_scene_note.argtypes = [ctxp]
        lib.rtx_scene_note.restype = C.c_char_p
        _librtx = lib
    return _librtx


def scene_kernel_source(scene) -> str:
    """rtx_scene_kernel_source: the scan of the scene as generated for its specialised kernel."""
    lib = load_librtx()
    ptr = scene.view_ptr if isinstance(scene, HostScene) else C.pointer(scene)
    needed = C.c_uint64(0)
    rc = lib.rtx_scene_kernel_source(ptr, None, 0, C.byref(needed))
    if rc != RTX_OK:
        raise RtxError(rc, lib.rtx_last_error().decode())
    buf = C.create_string_buffer(needed.value)
    rc = lib.rtx_scene_kernel_source(ptr, buf, needed.value, None)
    if rc != RTX_OK:
        raise RtxError(rc, lib.rtx_last_error().decode())
    return buf.value.decode()


def prepare_scene(scene) -> None:
    """rtx_prepare_scene: compile the scene's specialised kernel into the process's cache (no GPU needed)."""
    lib = load_librtx()
    ptr = scene.view_ptr if isinstance(scene, HostScene) else C.pointer(scene)
    rc = lib.rtx_prepare_scene(ptr)
    if rc != RTX_OK:
        raise RtxError(rc, lib.rtx_last_error().decode())


def load_libhost() -> C.CDLL:
    """Load the host scene API library (which links librtx.so)."""
    global _libhost
    if _libhost is None:
        load_librtx()
        path = libhost_path()
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: build it with `make -C raytrace_amd/host`")
        lib = C.CDLL(path)
        lib.rtxh_last_error.restype = C.c_char_p
        lib.rtxh_scene_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
        lib.rtxh_scene_create.restype = C.c_void_p
        lib.rtxh_scene_view.argtypes = [C.c_void_p]
        lib.rtxh_scene_view.restype = C.POINTER(RtxScene)
        lib.rtxh_scene_destroy.argtypes = [C.c_void_p]
        lib.rtxh_scene_destroy.restype = None
        lib.rtxh_image_write.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        lib.rtxh_image_write.restype = None
        lib.rtxh_resolve.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.rtxh_resolve.restype = None
        lib.rtxh_write_bmp.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        lib.rtxh_write_bmp.restype = C.c_int
        lib.rtxh_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        lib.rtxh_load_image.restype = C.c_int
        lib.rtxh_render_to_bmp.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_uint64, C.c_int, C.POINTER(C.c_ulonglong),
                                           C.POINTER(C.c_double)]
        lib.rtxh_render_to_bmp.restype = C.c_int
        lib.rtxh_set_kernel_compilation.argtypes = [C.c_int]
        lib.rtxh_set_kernel_compilation.restype = C.c_int
        _libhost = lib
    return _libhost


def _as_np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class HostScene:
    """A named scene built through the host scene API and flattened to RtxScene."""

    def __init__(self, name: str, width: int, height: int):
        lib = load_libhost()
        self._lib = lib
        self.name, self.width, self.height = name, width, height
        self._handle = lib.rtxh_scene_create(name.encode(), width, height)
        if not self._handle:
            raise ValueError(lib.rtxh_last_error().decode())
        self.view_ptr = lib.rtxh_scene_view(self._handle)

    @property
    def view(self) -> RtxScene:
        return self.view_ptr.contents

    def close(self):
        if self._handle:
            self._lib.rtxh_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def arrays(self) -> dict:
        """Copy of every array of the flattened scene, for inspection and fixtures."""
        s = self.view
        out = {}
        n_le = s.n_light_entries or s.n_lights  # all light entries, nested lists included
        for name, n in (("prim_kind", s.n_prims), ("prim_flip", s.n_prims), ("prim_material", s.n_prims),
                        ("prim_chain", s.n_prims), ("chain_begin", s.n_chains + 1), ("step_kind", s.n_steps),
                        ("mat_kind", s.n_materials), ("mat_texture", s.n_materials),
                        ("tex_kind", s.n_textures), ("tex_odd", s.n_textures), ("tex_even", s.n_textures),
                        ("tex_offset", s.n_textures),
                        ("light_kind", n_le), ("light_first", n_le), ("light_count", n_le),
                        ("light_chain", n_le)):
            out[name] = _as_np(getattr(s, name), n, np.uint32) if getattr(s, name) else np.zeros(n, np.uint32)
        for name, n in (("prim_a", s.n_prims), ("prim_b", s.n_prims), ("prim_c", s.n_prims),
                        ("prim_d", s.n_prims), ("prim_e", s.n_prims), ("step_x", s.n_steps),
                        ("step_y", s.n_steps), ("step_z", s.n_steps), ("step_w", s.n_steps),
                        ("mat_param", s.n_materials), ("tex_r", s.n_textures), ("tex_g", s.n_textures),
                        ("tex_b", s.n_textures), ("light_a", n_le), ("light_b", n_le),
                        ("light_c", n_le), ("light_d", n_le), ("light_e", n_le)):
            out[name] = _as_np(getattr(s, name), n, np.float32)
        for name in ("cam_origin", "cam_ll", "cam_u", "cam_v", "background"):
            out[name] = np.array(list(getattr(s, name)), np.float32)
        out["light_is_list"] = np.array([s.light_is_list], np.uint32)
        out["n_lights"] = np.array([s.n_lights], np.uint32)
        out["light_sampling"] = np.array([s.light_sampling], np.uint32)
        out["background_mode"] = np.array([s.background_mode], np.uint32)
        out["texels"] = _as_np(s.texels, int(s.n_texel_bytes), np.uint8)
        return out


class Context:
    """One GPU context (``RtxContext``)."""

    def __init__(self, device: int = 0):
        self._lib = load_librtx()
        self._ctx = C.c_void_p()
        rc = self._lib.rtx_create(device, C.byref(self._ctx))
        if rc != RTX_OK:
            raise RtxError(rc, self._lib.rtx_last_error().decode())
        self.device = device

    def _check(self, rc: int):
        if rc != RTX_OK:
            raise RtxError(rc, self._lib.rtx_last_error().decode())

    def close(self):
        if self._ctx:
            self._lib.rtx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene, accel: int = RTX_ACCEL_AUTO):
        ptr = scene.view_ptr if isinstance(scene, HostScene) else C.pointer(scene)
        self._check(self._lib.rtx_upload_scene(self._ctx, ptr, accel))

    def set_option(self, option: int, value: int):
        self._check(self._lib.rtx_set_option(self._ctx, option, value))

    def note(self) -> str:
        """Why the uploaded scene's scan was not specialised ("" when it was or is no candidate)."""
        return self._lib.rtx_scene_note(self._ctx).decode()

    def info(self) -> RtxSceneInfo:
        out = RtxSceneInfo()
        self._check(self._lib.rtx_scene_info(self._ctx, C.byref(out)))
        return out

    @staticmethod
    def params(width, height, spp_begin, spp_count, mode=RTX_MODE_FAST, seed=0, max_depth=50,
               accumulate=False) -> RtxParams:
        return RtxParams(width, height, spp_begin, spp_count, max_depth, mode, seed,
                         1 if accumulate else 0, 0)

    def render(self, params: RtxParams, accum: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int]:
        """rtx_render: returns (sum image [h, w, 3] with row 0 = top, rays)."""
        if accum is None:
            accum = np.zeros((params.height, params.width, 3), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous
        assert accum.shape == (params.height, params.width, 3)
        rays = C.c_uint64(0)
        self._check(self._lib.rtx_render(self._ctx, C.byref(params), accum.ctypes.data, C.byref(rays)))
        return accum, rays.value

    def render_noalloc(self, params: RtxParams, accum: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int]:
        """rtx_render_noalloc: rtx_render over the library's static device buffer."""
        if accum is None:
            accum = np.zeros((params.height, params.width, 3), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous
        assert accum.shape == (params.height, params.width, 3)
        rays = C.c_uint64(0)
        self._check(self._lib.rtx_render_noalloc(self._ctx, C.byref(params), accum.ctypes.data, C.byref(rays)))
        return accum, rays.value

    def render_device(self, params: RtxParams, d_accum: int, d_rays: int = 0, stream: int = 0):
        """rtx_render_device on raw device pointers (e.g. torch tensor data_ptr())."""
        self._check(self._lib.rtx_render_device(self._ctx, C.byref(params), C.c_void_p(d_accum),
                                                C.c_void_p(d_rays) if d_rays else None,
                                                C.c_void_p(stream) if stream else None))

    def render_samples(self, params: RtxParams) -> Tuple[np.ndarray, np.ndarray]:
        """Per-sample radiance [h*w, spp, 3] (pixel = j*w+i, j from the bottom) and segment counts."""
        n = params.width * params.height
        rgb = np.zeros((n, params.spp_count, 3), np.float32)
        seg = np.zeros((n, params.spp_count), np.uint32)
        self._check(self._lib.rtx_render_samples(self._ctx, C.byref(params), rgb.ctypes.data, seg.ctypes.data))
        return rgb, seg

    def hit(self, origins: np.ndarray, directions: np.ndarray, t0: float = 0.001,
            t1: float = 3.402823466e+38, mode: int = RTX_MODE_DETERMINISTIC) -> np.ndarray:
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        hits = np.zeros(o.shape[0], HIT_DTYPE)
        self._check(self._lib.rtx_hit(self._ctx, o.shape[0], o.ctypes.data, d.ctypes.data, t0, t1, mode,
                                      hits.ctypes.data))
        return hits

    def resolve_device(self, width: int, height: int, ns: int, d_accum: int, stream: int = 0,
                       out: Optional[np.ndarray] = None) -> np.ndarray:
        """rtx_resolve_device into `out` (a C-contiguous uint8 [height, width, 3] array; give one that
        lives in page-locked memory and the copy back runs on a DMA engine instead of through a staging
        buffer filled by copy kernels) or into a new array."""
        if out is None:
            out = np.zeros((height, width, 3), np.uint8)
        assert out.dtype == np.uint8 and out.flags.c_contiguous and out.shape == (height, width, 3)
        self._check(self._lib.rtx_resolve_device(self._ctx, width, height, ns, C.c_void_p(d_accum),
                                                 out.ctypes.data, C.c_void_p(stream) if stream else None))
        return out


    def resolve_to_device(self, width: int, height: int, ns: int, d_accum: int, d_rgb8: int, stream: int = 0) -> None:
        """rtx_resolve_to_device: the 8-bit image into the device buffer at `d_rgb8`; enqueues and returns."""
        self._check(self._lib.rtx_resolve_to_device(self._ctx, width, height, ns, C.c_void_p(d_accum), C.c_void_p(d_rgb8),
                                                    C.c_void_p(stream) if stream else None))


def render_multi(contexts, params: RtxParams, accum: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int]:
    """rtx_render_multi: one frame split by sample range over `contexts` (each holding the same
    scene), summed on the first context's GPU; returns (sum image, rays)."""
    lib = load_librtx()
    if accum is None:
        accum = np.zeros((params.height, params.width, 3), np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous
    assert accum.shape == (params.height, params.width, 3)
    handles = (C.c_void_p * len(contexts))(*[c._ctx for c in contexts])
    rays = C.c_uint64(0)
    rc = lib.rtx_render_multi(handles, len(contexts), C.byref(params), accum.ctypes.data, C.byref(rays))
    if rc != RTX_OK:
        raise RtxError(rc, lib.rtx_last_error().decode())
    return accum, rays.value


def device_count() -> int:
    return load_librtx().rtx_device_count()


def image_write(rgb: np.ndarray) -> np.ndarray:
    """Host Image::write filter chain for an [n, 3] fp32 array."""
    lib = load_libhost()
    a = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), np.uint8)
    lib.rtxh_image_write(a.shape[0], a.ctypes.data, out.ctypes.data)
    return out


def resolve_host(sum_rgb: np.ndarray, ns: int) -> np.ndarray:
    lib = load_libhost()
    a = np.ascontiguousarray(sum_rgb, np.float32)
    h, w, _ = a.shape
    out = np.zeros((h, w, 3), np.uint8)
    lib.rtxh_resolve(w, h, ns, a.ctypes.data, out.ctypes.data)
    return out


def load_image(path: str) -> np.ndarray:
    """The texels ImageTexture(path) holds: (h, w, 3) uint8, rows top-down."""
    lib = load_libhost()
    w, h = C.c_int(0), C.c_int(0)
    if not lib.rtxh_load_image(str(path).encode(), C.byref(w), C.byref(h), None, 0):
        raise ValueError(lib.rtxh_last_error().decode())
    out = np.zeros((h.value, w.value, 3), np.uint8)
    lib.rtxh_load_image(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes)
    return out


def write_bmp(path: str, rgb8: np.ndarray) -> None:
    lib = load_libhost()
    a = np.ascontiguousarray(rgb8, np.uint8)
    h, w, _ = a.shape
    if not lib.rtxh_write_bmp(path.encode(), w, h, a.ctypes.data):
        raise OSError(f"cannot write {path}")
