"""The oracle against the golden vectors (CPU only).

The golden vectors were produced by the reference tracer itself (the unmodified
reference sources compiled by oracle/Makefile, see tests/golden/make_golden.py).
The CPU restatement (oracle/restate.cpp) keeps the reference's operation order, so it
must reproduce them BIT FOR BIT; where the reference build is present in this
checkout it is re-checked against its own fixtures too.
"""
import numpy as np
import pytest

from oracle import ref, restate
from raytrace_amd import api
from tests import util


@pytest.mark.parametrize("key,meta", util.sample_cases())
def test_restatement_reproduces_reference_samples(key, meta):
    g = util.load_golden(key)
    scene = api.HostScene(meta["scene"], meta["w"], meta["h"])
    out = restate.render(scene, meta["w"], meta["h"], meta["spp_begin"], meta["spp"], seed=meta["seed"], threads=4,
                         per_sample=True)
    util.assert_samples_match(out["samples"], out["segments"], g["samples"], g["segments"], exact=True)
    assert out["rays"] == int(g["rays"]) == meta["rays"]
    assert np.array_equal(out["sum"], g["sum"], equal_nan=True)


@pytest.mark.parametrize("name", ["cornell", "cornell_extra", "random_field", "bvh_stress_500", "textured"])
def test_restatement_reproduces_reference_hits(name):
    g = util.load_golden("hits_" + name)
    scene = api.HostScene(name, 32, 32)
    got = restate.hit(scene, g["origins"], g["directions"])
    assert np.array_equal(got["hit"], g["hit"].astype(np.int32))
    m = g["hit"] == 1
    assert m.sum() > 1000
    for field in ("t", "p", "n", "uv"):  # uv: Rect.cpp:27-28, Util.h:78-83
        assert np.array_equal(got[field][m], g[field][m]), field
    # material kind of the winner
    kinds = scene.arrays()["mat_kind"]
    assert np.array_equal(kinds[got["material"][m]].astype(np.int8), g["mat_kind"][m])


@pytest.mark.parametrize("key,meta", [c for c in util.sample_cases()
                                      if c[1]["scene"] in ("cornell_extra", "random_field", "bvh_stress_500", "textured")])
def test_restatement_bvh_reproduces_reference_samples(key, meta):
    """The restatement's own hierarchy (built in oracle/restate.cpp, explicit tie rule) against
    the reference's O(N) scan: the same paths bit for bit."""
    g = util.load_golden(key)
    scene = api.HostScene(meta["scene"], meta["w"], meta["h"])
    out = restate.render(scene, meta["w"], meta["h"], meta["spp_begin"], meta["spp"], seed=meta["seed"], threads=4,
                         per_sample=True, bvh=True)
    util.assert_samples_match(out["samples"], out["segments"], g["samples"], g["segments"], exact=True)
    assert out["rays"] == meta["rays"]


@pytest.mark.parametrize("name", ["cornell", "cornell_extra", "random_field", "bvh_stress_500", "textured"])
def test_restatement_bvh_reproduces_reference_hits(name):
    g = util.load_golden("hits_" + name)
    scene = api.HostScene(name, 32, 32)
    got = restate.hit(scene, g["origins"], g["directions"], bvh=True)
    assert np.array_equal(got["hit"], g["hit"].astype(np.int32))
    m = g["hit"] == 1
    for field in ("t", "p", "n", "uv"):
        assert np.array_equal(got[field][m], g[field][m]), field


def test_bvh_flop_count_is_far_below_the_scan():
    scene = api.HostScene("random_field", 48, 27)
    lin = restate.count_flops(scene, 48, 27, 0, 2, seed=2, threads=4)
    bvh = restate.count_flops(scene, 48, 27, 0, 2, seed=2, threads=4, bvh=True)
    assert lin["rays"] == bvh["rays"]
    assert bvh["flops"] < 0.2 * lin["flops"]


def test_restatement_sky_and_cosine_pdf_match_reference():
    """Scene::backgroundSky and CosinePdf, the two reference functions the sky / no-light
    extension consists of (the reference's color() itself can render neither)."""
    g = util.load_golden("sky_and_cosine_pdf")
    assert np.array_equal(restate.background_sky(g["sky_dirs"]), g["sky"])
    dirs, vals = restate.cosine_pdf(g["normals"], seed=int(g["seed"]))
    assert np.array_equal(dirs, g["cos_dirs"])
    assert np.array_equal(vals, g["cos_values"])
    assert (vals > 0).all()


def test_restatement_light_shape_density_and_sampling_match_reference():
    """Shape::pdf_value / Shape::random of whole light shapes (ShapeList.cpp:19-35, Rect.cpp:36-68,
    Sphere.cpp:38-58, Shape.h:8-9), function by function: a flat list, one sphere, nested lists
    with shapes of the default density, a list of 40 lights; and, for general light sampling, the
    reference's ShapeList over the general light shapes of oracle/ref_driver.cpp (a flipped YZ rect,
    a flipped XY rect, a rotated and translated box as the list of its faces)."""
    g = util.load_golden("light_pdf")
    for name in ("cornell", "small", "cornell_light_tree", "cornell_many_lights", "cornell_side_lights"):
        scene = api.HostScene(name, 32, 32)
        vals, dirs = restate.light_pdf(scene, g[name + "_o"], g[name + "_v"], seed=int(g["seed"]))
        assert np.array_equal(vals, g[name + "_values"], equal_nan=True), name
        assert np.array_equal(dirs, g[name + "_dirs"], equal_nan=True), name
        assert (g[name + "_values"] > 0).sum() > 200, name


def test_restatement_general_light_sampling_is_unbiased():
    """General light sampling (rects of every axis, rotated / translated / boxed lights) is an
    extension of the reference's leaf densities (its paths are pinned sample for sample through the
    reference's own integrator, see the golden "samples_cornell_side_lights"); what the extension
    itself owes is that it leaves the expectation of the estimator alone.  Against the same scene
    rendered with the material's pdf only."""
    w, h = 64, 36
    lit = restate.render(api.HostScene("cornell_side_lights", w, h), w, h, 0, 1024, seed=1, threads=8)["sum"] / 1024
    unlit = restate.render(api.HostScene("cornell_side_lights_nolight", w, h), w, h, 0, 4096, seed=2, threads=8)["sum"] / 4096
    assert abs(lit.mean() / unlit.mean() - 1) < 0.01

    def blocks(x):
        return x.reshape(6, 6, 8, 8, 3).mean(axis=(1, 3))

    assert np.abs(blocks(lit) - blocks(unlit)).max() < 0.15 * blocks(unlit).mean()


def test_restatement_image_write_matches_reference():
    g = util.load_golden("image_write")
    assert np.array_equal(restate.image_write(g["rgb"]), g["rgb8"])


def test_restatement_thread_count_does_not_matter():
    scene = api.HostScene("cornell_extra", 40, 24)
    a = restate.render(scene, 40, 24, 3, 5, seed=1, threads=1, per_sample=True)
    b = restate.render(scene, 40, 24, 3, 5, seed=1, threads=7, per_sample=True)
    assert np.array_equal(a["samples"], b["samples"], equal_nan=True) and a["rays"] == b["rays"]


def test_restatement_sample_ranges_compose():
    """Streams are keyed by the absolute sample index: [0,6) == [0,2) + [2,6) sample for sample."""
    scene = api.HostScene("small", 32, 18)
    whole = restate.render(scene, 32, 18, 0, 6, seed=5, threads=4, per_sample=True)
    lo = restate.render(scene, 32, 18, 0, 2, seed=5, threads=4, per_sample=True)
    hi = restate.render(scene, 32, 18, 2, 4, seed=5, threads=4, per_sample=True)
    assert np.array_equal(whole["samples"][:, :2], lo["samples"])
    assert np.array_equal(whole["samples"][:, 2:], hi["samples"])
    assert whole["rays"] == lo["rays"] + hi["rays"]


def test_flop_count_is_stable():
    """The op-counting build walks the same paths as the float build."""
    scene = api.HostScene("cornell", 48, 27)
    c = restate.count_flops(scene, 48, 27, 0, 4, seed=2, threads=3)
    r = restate.render(scene, 48, 27, 0, 4, seed=2, threads=3)
    assert c["rays"] == r["rays"]
    assert 150 < c["flops"] / c["rays"] < 600


needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built in this checkout")


@needs_ref
@pytest.mark.parametrize("key,meta", util.sample_cases()[:3])
def test_reference_build_reproduces_its_fixtures(key, meta):
    g = util.load_golden(key)
    out = ref.RefScene(meta["scene"], meta["w"], meta["h"]).render(meta["spp_begin"], meta["spp"], seed=meta["seed"],
                                                                  threads=4, per_sample=True)
    util.assert_samples_match(out["samples"], out["segments"], g["samples"], g["segments"], exact=True)


@needs_ref
def test_builtin_scene_equals_transcribed_scene():
    """scenes.inc's "cornell" is the scene of the reference's own Scene::build()."""
    a = ref.RefScene("cornell_builtin", 40, 20).render(0, 4, seed=3, threads=4, per_sample=True)
    b = ref.RefScene("cornell", 40, 20).render(0, 4, seed=3, threads=4, per_sample=True)
    assert np.array_equal(a["samples"], b["samples"], equal_nan=True)
    assert np.array_equal(a["segments"], b["segments"])


@needs_ref
def test_reference_and_restatement_agree_on_fresh_inputs():
    for name, w, h, spp, seed in (("cornell_extra", 36, 20, 6, 1234), ("random_field", 30, 16, 3, 99),
                                  ("textured", 40, 22, 5, 31), ("cornell_light_tree", 40, 22, 6, 5),
                                  ("cornell_many_lights", 40, 22, 6, 5), ("cornell_side_lights", 40, 22, 6, 5)):
        scene = api.HostScene(name, w, h)
        a = restate.render(scene, w, h, 10, spp, seed=seed, threads=4, per_sample=True)
        b = ref.RefScene(name, w, h).render(10, spp, seed=seed, threads=4, per_sample=True)
        util.assert_samples_match(a["samples"], a["segments"], b["samples"], b["segments"], exact=True)


@needs_ref
def test_light_shape_golden_is_what_the_reference_returns():
    g = util.load_golden("light_pdf")
    for name in ("cornell", "cornell_many_lights", "cornell_side_lights"):
        vals, dirs = ref.RefScene(name, 32, 32).light_pdf(g[name + "_o"], g[name + "_v"], seed=int(g["seed"]))
        assert np.array_equal(vals, g[name + "_values"], equal_nan=True)
        assert np.array_equal(dirs, g[name + "_dirs"], equal_nan=True)


def test_resolve_threshold_table_is_the_references_byte_staircase():
    """The 255 thresholds the device resolve searches (generated from the reference's Image::write,
    exhaustively checked on [0, 1] when generated): the compiled-in table is the fixture, and
    the fixture reproduces the reference's bytes - the committed Image::write table, and live
    through the reference / the host filter chain on random values and on every threshold's
    neighbours."""
    import os
    import re

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = util.load_golden("resolve_thresholds")
    T = g["threshold_bits"].astype(np.uint32)
    assert int(g["exhaustive_floats_checked"]) == 0x3f800001  # every float in [0, 1]
    inc = open(os.path.join(ROOT, "raytrace_amd", "csrc", "rtx_resolve_table.inc")).read()
    compiled = np.array([int(v, 16) for v in re.findall(r"0x([0-9a-f]{8})u", inc)], np.uint32)
    assert np.array_equal(compiled, T)
    edges = T[1:].view(np.float32)

    def staircase(x):
        x = np.asarray(x, np.float32)
        r = np.where(np.isnan(x[:, 0]), np.float32(1), x[:, 0])  # DenanFilter.h:8-13
        gch = np.where(np.isnan(x[:, 1]), np.float32(0), x[:, 1])
        b = np.where(np.isnan(x[:, 2]), np.float32(1), x[:, 2])
        # (powf(-inf, 1/2.2) is +inf: the one negative input whose code is not 0)
        return np.stack([np.searchsorted(edges, np.where(np.isneginf(c), np.float32(np.inf), c), side="right")
                         for c in (r, gch, b)], 1).astype(np.uint8)

    iw = util.load_golden("image_write")
    assert np.array_equal(staircase(iw["rgb"]), iw["rgb8"])
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.random((20000, 3)).astype(np.float32) ** 3,
                        np.repeat(np.concatenate([T[1:] - 1, T[1:], T[1:] + 1]).astype(np.uint32).view(np.float32)[:, None], 3, 1),
                        rng.normal(0, 2, (2000, 3)).astype(np.float32)])
    assert np.array_equal(staircase(x), api.image_write(x))  # the product's host chain
    if ref.available():
        assert np.array_equal(staircase(x), ref.image_write(x))


def test_material_scatter_functions_match_the_reference():
    """Material::scatter / emitted / scattering_pdf function by function (Lambertian.cpp:16-25,
    Metal.cpp:9-17 incl. the g++ draw order of the fuzz vector, Dielectric.cpp:7-45, DiffuseLight.h:17-28):
    the restatement's material functions - the ones its color() calls - return the bits of the
    reference's classes on 1 024 (direction, normal) pairs per material, fed the same streams."""
    from tests.golden.make_golden import SCATTER_MATERIALS

    g = util.load_golden("scatter")
    for name, kind, param, tex in SCATTER_MATERIALS:
        got = restate.scatter(kind, param, tex, g["d"], g["n"], g["scattered"], seed=int(g["seed"]))
        for k in ("ok", "specular"):
            assert np.array_equal(got[k], g[f"{name}_{k}"].astype(np.int32)), (name, k)
        for k in ("dir", "albedo", "emitted", "spdf"):
            assert np.array_equal(got[k], g[f"{name}_{k}"], equal_nan=True), (name, k)
        if ref.available():  # the fixture is what the reference returns today
            live = ref.scatter(kind, param, tex, g["d"], g["n"], g["scattered"], seed=int(g["seed"]))
            for k in ("dir", "albedo", "emitted", "spdf"):
                assert np.array_equal(live[k], g[f"{name}_{k}"], equal_nan=True), (name, k)
    # the vectors exercise what they are meant to: both outcomes of Metal's rejection, both branches
    # of the dielectric (reflection and refraction), emission on the front face only
    assert 0 < g["metal_fuzz03_ok"].sum() < g["metal_fuzz03_ok"].size
    assert 0 < g["diffuse_light_emitted"].any(axis=1).sum() < g["d"].shape[0]
    assert g["diffuse_light_ok"].sum() == 0 and g["lambertian_ok"].all() and g["dielectric15_ok"].all()
    assert (g["lambertian_spdf"] > 0).any() and (g["lambertian_spdf"] == 0).any()


def test_jpeg_fixtures_are_what_the_references_loader_returns(tmp_path):
    """jpeg_cases.npz against the live reference build: the stored texels are stbi_load's."""
    if not ref.available():
        pytest.skip("oracle/_ref is not built")
    g = util.load_golden("jpeg_cases")
    names = sorted(k[:-5] for k in g.files if k.endswith("_file"))
    assert "asset_field001" in names and len(names) >= 20
    for name in names:
        path = tmp_path / (name + ".jpg")
        path.write_bytes(g[name + "_file"].tobytes())
        got = ref.load_image(path)
        assert got is not None and np.array_equal(got, g[name + "_texels"]), name


@pytest.mark.parametrize("name,w,h,spp", [("bvh_stress_500", 32, 18, 3), ("random_field", 32, 18, 2), ("cornell_extra", 32, 18, 4)])
def test_the_renderers_tree_walked_on_the_host_finds_the_scans_hits(name, w, h, spp):
    """The tree of the renderer's own builder (raytrace_amd/csrc/rtx_bvh.cpp, exported by
    tests/tree_export.cpp with its 16-bit boxes decoded) walked by the CPU restatement in the GPU
    kernel's order: the tree only prunes ShapeList::hit's scan (ShapeList.cpp:5-17), so every sample
    must come out bit for bit as from the scan.  This is the walk whose operations price a ray of the
    BVH workloads (tests/golden/flops_per_ray.json, keys ":bvh_product")."""
    scene = api.HostScene(name, w, h)
    tree = util.product_tree(scene)
    assert len(tree["item_begin"]) >= 5 and tree["depth"] >= 2
    a = restate.render(scene, w, h, 0, spp, seed=5, threads=4, per_sample=True)
    b = restate.render(scene, w, h, 0, spp, seed=5, threads=4, per_sample=True, tree=tree)
    assert np.array_equal(a["segments"], b["segments"])
    assert np.array_equal(a["samples"], b["samples"], equal_nan=True)


def test_flops_fixture_prices_tree_scenes_with_the_renderers_tree():
    import json
    import os

    table = json.load(open(os.path.join(util.GOLDEN, "flops_per_ray.json")))
    for scene in ("random_field", "bvh_stress"):
        own, product = table[scene + ":bvh"], table[scene + ":bvh_product"]
        assert product["rays"] == own["rays"]  # the same rays
        assert 100 < product["flops_per_ray"] < own["flops_per_ray"]  # near-first order over surface-area splits visits less
        assert product["inner_nodes_per_ray"] * 24 < product["flops_per_ray"]
    # a small case live: the counted operations are those of the fixture's method
    scene = api.HostScene("random_field", 48, 27)
    c = restate.count_flops(scene, 48, 27, 0, 1, seed=0, threads=4, tree=util.product_tree(scene))
    assert abs(c["flops"] / c["rays"] / table["random_field:bvh_product"]["flops_per_ray"] - 1) < 0.1
    assert abs(c["tree_nodes"] / c["rays"] / table["random_field:bvh_product"]["inner_nodes_per_ray"] - 1) < 0.1
