"""ctypes wrapper of oracle/_ref/libref_oracle.so: the UNMODIFIED reference tracer
(compiled in place from /root/reference by oracle/Makefile) behind the C driver
oracle/ref_driver.cpp.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_oracle.so")
_lib = None


def available() -> bool:
    return os.path.exists(LIB_PATH)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        lib = C.CDLL(LIB_PATH)
        lib.ref_scene_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
        lib.ref_scene_create.restype = C.c_void_p
        lib.ref_scene_destroy.argtypes = [C.c_void_p]
        lib.ref_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.ref_render.restype = C.c_int
        lib.ref_hit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p] * 6
        lib.ref_hit.restype = C.c_int
        lib.ref_camera_rays.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_image_write.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        lib.ref_resolve.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ref_write_bmp.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        lib.ref_write_bmp.restype = C.c_int
        lib.ref_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        lib.ref_load_image.restype = C.c_int
        lib.ref_write_jpg.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.ref_write_jpg.restype = C.c_int
        _lib = lib
    return _lib


class RefScene:
    def __init__(self, name: str, width: int, height: int):
        self.lib = load()
        self.w, self.h = width, height
        self.handle = self.lib.ref_scene_create(name.encode(), width, height)
        if not self.handle:
            raise ValueError(f"unknown scene {name!r}")

    def close(self):
        if self.handle:
            self.lib.ref_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, spp_begin: int, spp_count: int, seed: int = 0, threads: int = 1, per_sample: bool = False):
        """Returns dict(sum [h,w,3] top-first, rays, draws[, samples [h*w,spp,3], segments [h*w,spp]])."""
        n = self.w * self.h
        acc = np.zeros((self.h, self.w, 3), np.float32)
        rays, draws = C.c_uint64(0), C.c_uint64(0)
        samples = np.zeros((n, spp_count, 3), np.float32) if per_sample else None
        segs = np.zeros((n, spp_count), np.uint32) if per_sample else None
        rc = self.lib.ref_render(self.handle, spp_begin, spp_count, seed, threads, acc.ctypes.data,
                                 samples.ctypes.data if per_sample else None,
                                 segs.ctypes.data if per_sample else None, C.byref(rays), C.byref(draws))
        assert rc == 0
        out = {"sum": acc, "rays": rays.value, "draws": draws.value}
        if per_sample:
            out["samples"], out["segments"] = samples, segs
        return out

    def hit(self, origins, directions, t0=0.001, t1=3.402823466e+38):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        n = o.shape[0]
        hit = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        p = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        uv = np.zeros((n, 2), np.float32)
        mk = np.zeros(n, np.int32)
        rc = self.lib.ref_hit(self.handle, n, o.ctypes.data, d.ctypes.data, t0, t1, hit.ctypes.data, t.ctypes.data,
                              p.ctypes.data, nrm.ctypes.data, uv.ctypes.data, mk.ctypes.data)
        assert rc == 0
        return {"hit": hit, "t": t, "p": p, "n": nrm, "uv": uv, "mat_kind": mk}

    def light_pdf(self, origins, directions, seed=0):
        """Shape::pdf_value(o, v) of the scene's light shape and Shape::random(o), entry k drawing
        from the stream of (pixel k, sample 0, seed).  Returns (values, directions)."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        vals = np.zeros(o.shape[0], np.float32)
        dirs = np.zeros_like(o)
        rc = self.lib.ref_light_pdf(C.c_void_p(self.handle), C.c_int(o.shape[0]), C.c_void_p(o.ctypes.data),
                                    C.c_void_p(d.ctypes.data), C.c_uint64(seed), C.c_void_p(vals.ctypes.data),
                                    C.c_void_p(dirs.ctypes.data))
        assert rc == 0
        return vals, dirs

    def camera_rays(self, uv):
        a = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        o = np.zeros((a.shape[0], 3), np.float32)
        d = np.zeros((a.shape[0], 3), np.float32)
        self.lib.ref_camera_rays(self.handle, a.shape[0], a.ctypes.data, o.ctypes.data, d.ctypes.data)
        return o, d


def image_write(rgb):
    lib = load()
    a = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), np.uint8)
    lib.ref_image_write(a.shape[0], a.ctypes.data, out.ctypes.data)
    return out


def resolve(sum_rgb, ns):
    lib = load()
    a = np.ascontiguousarray(sum_rgb, np.float32)
    h, w, _ = a.shape
    out = np.zeros((h, w, 3), np.uint8)
    lib.ref_resolve(w, h, ns, a.ctypes.data, out.ctypes.data)
    return out


def write_bmp(path, rgb8):
    lib = load()
    a = np.ascontiguousarray(rgb8, np.uint8)
    h, w, _ = a.shape
    return lib.ref_write_bmp(path.encode(), w, h, a.ctypes.data)


def load_image(path):
    """The texels the reference's ImageTexture holds for `path` (stb_image, ImageTexture.h:9):
    (h, w, channels) uint8, or None if the file cannot be read."""
    lib = load()
    w, h = C.c_int(0), C.c_int(0)
    n = lib.ref_load_image(str(path).encode(), C.byref(w), C.byref(h), None, 0)
    if n == 0:
        return None
    out = np.zeros((h.value, w.value, n), np.uint8)
    lib.ref_load_image(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes)
    return out


def write_jpg(path, pixels, quality):
    """A JPEG of an (h, w, 1 | 3) uint8 image through the encoder the reference vendors."""
    lib = load()
    a = np.ascontiguousarray(pixels, np.uint8)
    h, w, n = a.shape
    return lib.ref_write_jpg(str(path).encode(), w, h, n, a.ctypes.data, int(quality))


def background_sky(dirs):
    """Scene::backgroundSky (Scene.h:24-28) of n directions."""
    lib = load()
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    out = np.zeros_like(d)
    lib.ref_background_sky(C.c_int(d.shape[0]), C.c_void_p(d.ctypes.data), C.c_void_p(out.ctypes.data))
    return out


def cosine_pdf(normals, seed=0):
    """CosinePdf::generate / value (CosinePdf.cpp:6-20) around n normals; entry k draws from the
    stream of (pixel k, sample 0, seed).  Returns (directions, values)."""
    lib = load()
    nn = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    dirs = np.zeros_like(nn)
    vals = np.zeros(nn.shape[0], np.float32)
    lib.ref_cosine_pdf(C.c_int(nn.shape[0]), C.c_void_p(nn.ctypes.data), C.c_uint64(seed),
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
    rc = lib.ref_scatter(C.c_uint32(kind), C.c_float(param), C.c_void_p(t.ctypes.data), C.c_int(n), C.c_void_p(d.ctypes.data),
                        C.c_void_p(nn.ctypes.data), C.c_void_p(sd.ctypes.data), C.c_uint64(seed),
                        *[C.c_void_p(out[k].ctypes.data) for k in ("ok", "specular", "dir", "albedo", "emitted", "spdf")])
    assert rc == 0
    return out
