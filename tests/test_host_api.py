"""Host scene API, flattening, image pipeline and ABI surface (CPU only, no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from raytrace_amd import api
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_librtx_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rtx.h")).read()
    declared = set(re.findall(r"\b(rtx_[a-z_]+)\s*\(", header))
    assert declared == set(api.RTX_SYMBOLS), declared ^ set(api.RTX_SYMBOLS)
    lib = ctypes.CDLL(api.librtx_path())
    for sym in declared:
        assert hasattr(lib, sym), sym
    host = ctypes.CDLL(api.libhost_path())
    for sym in api.RTXH_SYMBOLS:
        assert hasattr(host, sym), sym


def test_struct_layout_matches_header():
    # 64-bit layout of include/rtx.h: catches a drifted ctypes mirror
    assert ctypes.sizeof(api.RtxParams) == 40
    assert ctypes.sizeof(api.RtxHit) == 48
    assert api.HIT_DTYPE.itemsize == 48
    assert ctypes.sizeof(api.RtxSceneInfo) == 32
    assert api.RtxScene.cam_origin.offset % 4 == 0


def test_struct_layout_matches_the_compiled_header(tmp_path):
    """sizeof / offsetof of every struct of include/rtx.h as gcc lays them out against the ctypes mirrors."""
    import subprocess

    structs = {"RtxScene": api.RtxScene, "RtxParams": api.RtxParams, "RtxHit": api.RtxHit,
               "RtxSceneInfo": api.RtxSceneInfo}
    lines = []
    for name, cls in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    src = tmp_path / "layout.c"
    src.write_text("#include <stdio.h>\n#include <stddef.h>\n#include \"rtx.h\"\nint main(void) {\n"
                   + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True,
                                                       text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for field, _ in cls._fields_:
            assert int(got[f"{name}.{field}"]) == getattr(cls, field).offset, (name, field)
    assert api.RTX_ABI_VERSION == int(re.search(r"#define RTX_ABI_VERSION (\d+)u", open(
        os.path.join(ROOT, "include", "rtx.h")).read()).group(1))


def test_scene_kernels_compile_without_a_gpu():
    """rtx_prepare_scene: the trace kernels around a small scene's scan, compiled by the HIP run-time
    compiler from the sources inside librtx.so (the code path an upload takes, minus the device);
    scenes the compiled variant does not cover are refused with the reason."""
    import time

    # (a light list, a sky background without lights, a light tree: the three kernel variants)
    for name in ("cornell", "small", "bvh_stress_20", "sky_spheres", "cornell_light_tree"):
        scene = api.HostScene(name, 32, 18)
        t = time.time()
        api.prepare_scene(scene)
        first = time.time() - t
        t = time.time()
        api.prepare_scene(scene)  # from the process's cache
        assert time.time() - t < 0.5 * max(first, 0.2)
    with pytest.raises(api.RtxError) as e:
        api.prepare_scene(api.HostScene("random_field", 32, 18))
    assert e.value.code == api.RTX_ERR_UNSUPPORTED and "specialised" in str(e.value)
    bad = api.RtxScene.from_buffer_copy(api.HostScene("cornell", 8, 8).view)
    bad.abi_version += 1
    with pytest.raises(api.RtxError) as e:
        api.prepare_scene(bad)
    assert e.value.code == api.RTX_ERR_INVALID_ARGUMENT


def test_scene_features_and_lights_in_the_generated_source():
    """What a scene's kernels are told about it: the feature set (bit values of kFeat* in csrc/rtx_core.h) and,
    for a light list without a tree, its entries."""
    textured, needs_uv, metal, fuzz, dielectric, lambertian, emitter, l_rect, l_sphere, l_other = (1 << k for k in range(10))

    def features(name):
        text = api.scene_kernel_source(api.HostScene(name, 16, 9))
        return int(re.search(r"#define RTX_SPECIAL_FEATURES 0x([0-9a-f]+)u", text).group(1), 16), text

    f, text = features("cornell")  # aluminium mirror, glass, white walls, lamp; lights: ceiling rect + sphere
    assert f == metal | dielectric | lambertian | emitter | l_rect | l_sphere
    assert text.count("RTX_SPECIAL_LIGHT(") == 2 and "RTX_SPECIAL_LIGHT_WEIGHT 0x3f000000u" in text  # 1 / 2
    f, _ = features("cornell_classic")  # two white boxes, one ceiling lamp
    assert f == lambertian | emitter | l_rect
    f, _ = features("textured")
    assert f & textured and f & needs_uv
    f, text = features("sky_spheres")  # no light shape at all
    assert not f & (l_rect | l_sphere | l_other | emitter) and "RTX_SPECIAL_LIGHTS" not in text
    f, text = features("cornell_light_tree")  # nested lists: the tree kernels, no literal list
    assert f & l_rect and f & l_sphere and f & l_other and "RTX_SPECIAL_LIGHTS" not in text
    f, _ = features("cornell_extra")
    assert f & fuzz and f & dielectric


def test_kernel_source_into_a_short_buffer_is_cut_and_terminated():
    lib = api.load_librtx()
    scene = api.HostScene("cornell", 16, 9)
    needed = ctypes.c_uint64(0)
    assert lib.rtx_scene_kernel_source(scene.view_ptr, None, 0, ctypes.byref(needed)) == api.RTX_OK
    whole = api.scene_kernel_source(scene)
    assert needed.value == len(whole) + 1
    buf = ctypes.create_string_buffer(b"\xff" * 64, 64)
    assert lib.rtx_scene_kernel_source(scene.view_ptr, buf, 24, None) == api.RTX_OK
    assert buf.raw[:23] == whole.encode()[:23] and buf.raw[23] == 0 and buf.raw[24] == 0xff
    assert lib.rtx_scene_kernel_source(None, buf, 24, None) == api.RTX_ERR_INVALID_ARGUMENT


def test_scenes_compile_concurrently_from_several_threads():
    """rtx_prepare_scene is what an application calls from worker threads: four scenes at once, then each
    again from the cache."""
    import threading
    import time

    names = ["cornell_classic", "cornell_extra", "sky_sphere_lit", "cornell_side_lights"]
    scenes = [api.HostScene(n, 24, 16) for n in names]
    errors = []

    def work(scene):
        try:
            api.prepare_scene(scene)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(sc,)) for sc in scenes]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors and not any(t.is_alive() for t in threads), errors
    t0 = time.time()
    for sc in scenes:
        api.prepare_scene(sc)
    assert time.time() - t0 < 1.0


def test_a_scan_that_does_not_compile_is_refused_with_the_compilers_message(tmp_path):
    """The failure path of the run-time compilation (tests/special_check.cpp, host only): rtx_upload_scene
    keeps the table-driven kernels when it returns false."""
    import subprocess

    csrc = os.path.join(ROOT, "raytrace_amd", "csrc")
    exe = str(tmp_path / "special_check")
    # (rtx_sources.cpp copies the device sources in by paths relative to its own directory)
    sources = str(tmp_path / "rtx_sources.o")
    subprocess.run(["g++", "-O1", "-c", "rtx_sources.cpp", "-o", sources], check=True, capture_output=True, cwd=csrc)
    subprocess.run(["hipcc", "-std=c++17", "-O1", "-x", "hip", "--offload-arch=gfx950", "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "special_check.cpp"), os.path.join(csrc, "rtx_special.cpp"), "-x", "none",
                    sources, "-ldl", "-o", exe], check=True, capture_output=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
    # compilations in the background run in the helper program beside the library (here: beside the test program);
    # the last job of the run is still compiling when the process ends, which must be a clean exit
    import shutil

    helper = os.path.join(ROOT, "raytrace_amd", "rtx_compile")
    assert os.path.exists(helper), "raytrace_amd/rtx_compile is not built (make -C raytrace_amd/csrc)"
    shutil.copy(helper, str(tmp_path / "rtx_compile"))
    out = subprocess.run([exe, "--jobs"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK jobs"), out.stdout + out.stderr


def test_the_compile_helper_answers_in_the_wire_format():
    """raytrace_amd/rtx_compile by itself: a scene's text in, lowered kernel names and a code object out;
    a malformed request and a text that does not compile come back as errors."""
    import subprocess

    helper = os.path.join(ROOT, "raytrace_amd", "rtx_compile")
    text = api.scene_kernel_source(api.HostScene("cornell_classic", 16, 9))
    out = subprocess.run([helper], input=b"0 1\n" + text.encode(), capture_output=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith(b"OK\n")
    lines = out.stdout.split(b"\n", 5)
    assert b"trace_kernel" in lines[1] and b"trace_kernel" in lines[2] and b"hit_kernel" in lines[3]
    assert int(lines[4]) == len(lines[5]) and lines[5][:4] == b"\x7fELF"
    out = subprocess.run([helper], input=b"nonsense", capture_output=True, timeout=60)
    assert out.stdout.startswith(b"ERR\n")
    out = subprocess.run([helper], input=b"0 0\n#define RTX_SPECIAL_SCAN this is not C++;\n", capture_output=True, timeout=300)
    assert out.returncode == 1 and out.stdout.startswith(b"ERR\n") and b"run-time compilation failed" in out.stdout


def test_generated_scan_lists_every_primitive_of_the_scene_once():
    """rtx_scene_kernel_source: the reference's Cornell box (Scene.cpp:43-86) as generated code - one
    line per record, the positions 0 .. n-1 in order, coordinates as the bits of the scene's floats."""
    scene = api.HostScene("cornell", 32, 18)
    text = api.scene_kernel_source(scene)
    a = scene.arrays()
    rects = re.findall(r"RTX_SPECIAL_RECT\((\d), (\d), (\d), RTX_SPECIAL_RANGE0?, RTX_SPECIAL_RANGE0?, (0x[0-9a-f]{8})u, (0x[0-9a-f]{8})u, (0x[0-9a-f]{8})u, (0x[0-9a-f]{8})u, (0x[0-9a-f]{8})u, (\d+)u\)", text)
    spheres = re.findall(r"RTX_SPECIAL_SPHERE\((0x[0-9a-f]{8})u, (0x[0-9a-f]{8})u, (0x[0-9a-f]{8})u, (0x[0-9a-f]{8})u, (\d+)u\)", text)
    boxes = re.findall(r"RTX_SPECIAL_BOX\(.*?, (\d+)u\)", text)
    kinds = a["prim_kind"]
    assert len(spheres) == (kinds == api.RTX_PRIM_SPHERE).sum()
    assert len(rects) + 6 * len(boxes) == (kinds != api.RTX_PRIM_SPHERE).sum()
    positions = sorted([int(r[-1]) for r in rects] + [int(s[-1]) for s in spheres] + [int(b) for b in boxes])
    expect, pos = [], 0
    for _ in range(len(rects) + len(spheres)):
        expect.append(pos)
        pos += 1
    for _ in boxes:
        expect.append(pos)
        pos += 6
    assert positions == expect
    as_float = lambda h: np.array([int(h, 16)], np.uint32).view(np.float32)[0]
    # the light of the reference's scene: Rect XZ 213..343 x 227..332 at y = 554 (Scene.cpp:63-67)
    assert any((as_float(r[3]), as_float(r[4]), as_float(r[5]), as_float(r[6]), as_float(r[7])) == (213.0, 343.0, 227.0, 332.0, 554.0)
               and r[:3] == ("0", "2", "1") for r in rects)
    radii = sorted(float(np.sqrt(as_float(s[3]))) for s in spheres)
    assert radii == sorted(float(v) for v in a["prim_d"][kinds == api.RTX_PRIM_SPHERE])
    assert text.count("RTX_SPECIAL_GROUP_") == text.count("    { \\\n")
    # the walls span 0 .. 555: their interval tests are the single unsigned comparison
    assert text.count("RTX_SPECIAL_RANGE0, RTX_SPECIAL_RANGE0") == 5


def test_no_gpu_means_a_loud_error_not_a_fallback():
    if api.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(api.RtxError) as e:
        api.Context(0)
    assert e.value.code == api.RTX_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_cornell_flattening():
    s = api.HostScene("cornell", 64, 36)
    a = s.arrays()
    # Scene.cpp:67-78 order: 6 room rects, 2 spheres, then the box's 6 rects (Box.cpp:15-28)
    assert a["prim_kind"].tolist() == [2, 2, 1, 1, 1, 0, 3, 3, 0, 0, 1, 1, 2, 2]
    assert a["prim_flip"].tolist() == [1, 0, 1, 1, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1]
    assert a["prim_chain"].tolist() == [0] * 8 + [1] * 6
    assert a["chain_begin"].tolist() == [0, 0, 2]
    # outermost wrapper first: translate (265, 0, 295), then rotate 15 degrees about y
    assert a["step_kind"].tolist() == [0, 1]
    assert (a["step_x"][0], a["step_y"][0], a["step_z"][0]) == (265.0, 0.0, 295.0)
    half = np.radians(15.0) / 2
    assert abs(a["step_y"][1] - np.sin(half)) < 1e-6 and abs(a["step_w"][1] - np.cos(half)) < 1e-6
    assert a["step_x"][1] == 0.0 and a["step_z"][1] == 0.0
    # box faces: +z at 165, -z at 0, +y at 330, -y at 0, +x at 165, -x at 0
    assert a["prim_e"][8:].tolist() == [165.0, 0.0, 330.0, 0.0, 165.0, 0.0]
    # light list: the ceiling rect and the radius-90 proxy sphere (Scene.cpp:84-85)
    assert a["light_kind"].tolist() == [1, 3] and a["light_is_list"][0] == 1
    assert a["light_d"][1] == 90.0
    assert a["background"].tolist() == [0.0, 0.0, 0.0]
    kinds = a["mat_kind"][a["prim_material"]]
    assert kinds.tolist() == [0, 0, 3, 0, 0, 0, 1, 2] + [0] * 6


def test_builtin_build_equals_scenes_inc_cornell():
    a = api.HostScene("cornell_builtin", 80, 40).arrays()
    b = api.HostScene("cornell", 80, 40).arrays()
    for k in a:
        if k.startswith(("prim_", "chain", "step", "light", "cam", "background")) and k != "prim_material":
            assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["mat_kind"][a["prim_material"]], b["mat_kind"][b["prim_material"]])


def test_camera_matches_reference_rays():
    g = util.load_golden("camera_rays")
    uv = g["uv"]
    for name, w, h in (("cornell", 64, 36), ("cornell", 800, 400), ("small", 400, 225), ("bvh_stress_500", 1920, 1080)):
        a = api.HostScene(name, w, h).arrays()
        d = ((a["cam_ll"] + a["cam_u"] * uv[:, :1]) + a["cam_v"] * uv[:, 1:]) - a["cam_origin"]
        assert np.array_equal(d.astype(np.float32), g[f"{name}_{w}x{h}_d"])
        assert np.array_equal(np.broadcast_to(a["cam_origin"], d.shape), g[f"{name}_{w}x{h}_o"])


def test_scene_sizes():
    assert api.HostScene("small", 8, 8).view.n_prims == 5
    rf = api.HostScene("random_field", 8, 8)
    assert 470 <= rf.view.n_prims <= 490 and rf.view.n_chains == 1
    assert 1 in rf.arrays()["tex_kind"]  # the checker ground
    bs = api.HostScene("bvh_stress_300", 8, 8)
    a = bs.arrays()
    n_boxes = (a["prim_chain"] > 0).sum() // 6
    assert bs.view.n_prims == 2 + (300 - n_boxes) + 6 * n_boxes
    assert bs.view.n_chains == 1 + n_boxes
    with pytest.raises(ValueError):
        api.HostScene("no_such_scene", 8, 8)


def test_image_write_matches_reference_fixture():
    g = util.load_golden("image_write")
    assert np.array_equal(api.image_write(g["rgb"]), g["rgb8"])
    # NaN -> magenta (DenanFilter.h:8-13)
    assert api.image_write(np.full((1, 3), np.nan, np.float32)).tolist() == [[255, 0, 255]]


def test_bmp_writer_matches_reference_bytes(tmp_path):
    g = util.load_golden("bmp_7x5")
    path = str(tmp_path / "x.bmp")
    api.write_bmp(path, g["rgb8"])
    assert np.array_equal(np.frombuffer(open(path, "rb").read(), np.uint8), g["bmp_bytes"])


# ---------------------------------------------------------------------------
# ImageTexture (ImageTexture.h:7-29): the host decoders against Pillow
# ---------------------------------------------------------------------------
def _textured_texels(path, monkeypatch):
    monkeypatch.setenv("RTX_TEXTURE", str(path))
    scene = api.HostScene("textured", 16, 9)
    a = scene.arrays()
    images = np.flatnonzero(a["tex_kind"] == api.RTX_TEX_IMAGE)
    assert images.size >= 1
    t = int(images[0])
    w, h, off = int(a["tex_odd"][t]), int(a["tex_even"][t]), int(a["tex_offset"][t])
    return a["texels"][off:off + w * h * 3].reshape(h, w, 3)


def test_image_texture_bmp_loader_matches_pillow(monkeypatch):
    from PIL import Image

    path = os.path.join(ROOT, "tests", "golden", "texture_64x32.bmp")
    want = np.asarray(Image.open(path).convert("RGB"))
    got = _textured_texels(path, monkeypatch)
    assert got.shape == want.shape == (32, 64, 3)
    assert np.array_equal(got, want)  # rows top-down, RGB: what stbi_load returns (ImageTexture.h:9)


def test_image_texture_ppm_and_padded_bmp(tmp_path, monkeypatch):
    from PIL import Image

    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)  # 7 * 3 = 21 bytes per row: BMP pads to 24
    for name in ("t.ppm", "t.bmp"):
        Image.fromarray(img).save(tmp_path / name)
        assert np.array_equal(_textured_texels(tmp_path / name, monkeypatch), img), name


def _jpeg_cases():
    g = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
    return g, sorted(k[:-5] for k in g.files if k.endswith("_file"))


@pytest.mark.parametrize("name", _jpeg_cases()[1])
def test_image_texture_jpeg_decoder_returns_the_references_texels(tmp_path, name):
    """The reference decodes its texture asset with the stb_image it vendors (ImageTexture.h:9); a
    JPEG decoder is free in its rounding, so the product's decoder has to reproduce THAT decoder's
    texels.  The fixture holds files (the reference's asset, files from the encoder it vendors,
    and the layouts of tests/golden/jpeg_cases.py: 4:2:2, 4:4:0, 4:1:1, restart intervals, one
    scan per component, 16-bit quantisers, grey scale, RGB component ids) and what the
    reference's loader returned for each."""
    g, _ = _jpeg_cases()
    path = tmp_path / (name + ".jpg")
    path.write_bytes(g[name + "_file"].tobytes())
    want = g[name + "_texels"]
    if want.shape[2] == 1:  # grey scale: the product hands the flattener RGB
        want = np.repeat(want, 3, axis=2)
    got = api.load_image(str(path))
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_jpeg_texture_reaches_the_flattened_scene(tmp_path, monkeypatch):
    g, _ = _jpeg_cases()
    path = tmp_path / "field001.jpg"
    path.write_bytes(g["asset_field001_file"].tobytes())
    got = _textured_texels(path, monkeypatch)  # through ImageTexture and the scene flattener
    assert got.shape == (160, 160, 3)
    assert np.array_equal(got, g["asset_field001_texels"])


def test_unsupported_and_damaged_jpeg_files_are_refused_with_a_reason(tmp_path):
    g, _ = _jpeg_cases()
    data = bytearray(g["s420_57x35_file"].tobytes())
    sof = data.find(b"\xFF\xC0")
    data[sof + 1] = 0xC2  # claims to be progressive
    (tmp_path / "p.jpg").write_bytes(bytes(data))
    with pytest.raises(ValueError, match="progressive"):
        api.load_image(str(tmp_path / "p.jpg"))
    whole = g["s420_57x35_file"].tobytes()
    (tmp_path / "cut.jpg").write_bytes(whole[:200])  # ends inside the tables
    with pytest.raises(ValueError):
        api.load_image(str(tmp_path / "cut.jpg"))
    (tmp_path / "cut2.jpg").write_bytes(whole[:-300])  # ends inside the scan: decodes what is there
    try:
        assert api.load_image(str(tmp_path / "cut2.jpg")).shape == (35, 57, 3)
    except ValueError:
        pass


def test_missing_texture_file_is_an_error(tmp_path, monkeypatch):
    monkeypatch.setenv("RTX_TEXTURE", str(tmp_path / "nope.bmp"))
    with pytest.raises(ValueError):
        api.HostScene("textured", 16, 9)
    # an unsupported format is reported by the flattener, not dereferenced
    (tmp_path / "junk.bmp").write_bytes(b"not an image at all")
    monkeypatch.setenv("RTX_TEXTURE", str(tmp_path / "junk.bmp"))
    with pytest.raises(ValueError):
        api.HostScene("textured", 16, 9)


def test_light_tree_flattening():
    """A ShapeList nested in the light list becomes an RTX_LIGHT_LIST entry whose children lie behind
    the top-level entries; wrapped shapes keep Shape's defaults (Shape.h:8-9) in reference sampling."""
    s = api.HostScene("cornell_light_tree", 32, 18)
    a = s.arrays()
    L, O = api.RTX_LIGHT_LIST, api.RTX_LIGHT_OTHER
    assert int(a["n_lights"][0]) == 5 and int(a["light_sampling"][0]) == api.RTX_LIGHTS_REFERENCE
    # top level: XZ rect, nested list, XY rect, translated sphere (default), box (default);
    # then the nested list's three entries, then the innermost list's one
    assert a["light_kind"].tolist() == [1, L, 0, O, O, 1, 3, L, 1]
    assert (int(a["light_first"][1]), int(a["light_count"][1])) == (5, 3)
    assert (int(a["light_first"][7]), int(a["light_count"][7])) == (8, 1)
    assert not a["light_chain"].any()


def test_general_light_flattening():
    """General light sampling: flips vanish, Translate / Rotate become the leaf's chain, a Box is the
    list of its six faces (Box.cpp:15-28)."""
    s = api.HostScene("cornell_side_lights", 32, 18)
    a = s.arrays()
    L = api.RTX_LIGHT_LIST
    assert int(a["light_sampling"][0]) == api.RTX_LIGHTS_GENERAL and int(a["n_lights"][0]) == 3
    assert a["light_kind"].tolist() == [2, 0, L, 0, 0, 1, 1, 2, 2]
    assert (int(a["light_first"][2]), int(a["light_count"][2])) == (3, 6)
    chains = a["light_chain"].tolist()
    assert chains[:3] == [0, 0, 0] and len(set(chains[3:])) == 1 and chains[3] != 0
    c = chains[3]
    steps = a["step_kind"][a["chain_begin"][c]:a["chain_begin"][c + 1]].tolist()
    assert steps == [0, 1]  # outermost first: the translation, then the rotation
    # the same scene without a light shape
    b = api.HostScene("cornell_side_lights_nolight", 32, 18).arrays()
    assert b["light_kind"].size == 0 and np.array_equal(a["prim_kind"], b["prim_kind"])


def test_scene_without_light_and_with_sky_background():
    """No light shape and Scene::backgroundSky (Scene.h:24-28): an extension of the flattened
    scene, since the reference's color() requires a light and never calls the gradient."""
    a = api.HostScene("sky_spheres", 32, 18).arrays()
    assert a["light_kind"].size == 0 and int(a["light_is_list"][0]) == 1
    assert int(a["background_mode"][0]) == api.RTX_BACKGROUND_SKY
    assert int(api.HostScene("small", 32, 18).arrays()["background_mode"][0]) == api.RTX_BACKGROUND_CONSTANT
    assert a["prim_kind"].size == 4 + 6  # four spheres and the six rects of the box


def test_rtx_render_tool_parses_its_arguments_and_fails_loudly_without_a_gpu(tmp_path):
    """raytrace_amd/rtx_render (the counterpart of main.cpp:7-20): usage errors exit with 2; on a
    machine without a GPU a render fails with an error message and exit code 1 - there is no CPU
    path to fall back to (on the GPU box the same call renders, see tests/test_host_e2e_gpu.py)."""
    import subprocess

    from raytrace_amd import api

    exe = os.path.join(ROOT, "raytrace_amd", "rtx_render")
    if not os.path.exists(exe):
        pytest.skip("raytrace_amd/rtx_render not built")
    if api.device_count() != 0:
        pytest.skip("starts child processes: only where this process cannot have initialised a GPU")
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([exe, "--devices", "x"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
    out = str(tmp_path / "o.bmp")
    r = subprocess.run([exe, "--scene", "cornell_builtin", "--width", "16", "--height", "9", "--spp", "1", "--out", out,
                        "--devices", "0,0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "render failed" in r.stderr and not os.path.exists(out)
