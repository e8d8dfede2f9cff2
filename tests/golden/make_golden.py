#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the reference tracer itself
(oracle/_ref/libref_oracle.so = the unmodified reference sources compiled by
oracle/Makefile, fed the counter-based stream of include/rtx_det.h).

Run from the repository root in a container that holds /root/reference:
    make -C oracle && python tests/golden/make_golden.py
The reference has no fixtures of its own for this path, so these files are what pins
the CPU restatement (and through it the GPU path) wherever the reference is absent.
Only data is stored: inputs (scene names, seeds, rays) and the reference's outputs.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref, restate  # noqa: E402
from raytrace_amd import api  # noqa: E402

SAMPLE_CASES = [  # scene, w, h, spp_begin, spp, seed
    ("cornell", 64, 36, 0, 16, 7),
    ("cornell", 24, 16, 100, 4, 12345678901234),
    ("cornell_classic", 32, 18, 0, 8, 7),
    ("cornell_extra", 32, 18, 0, 8, 7),
    ("small", 32, 18, 0, 8, 7),
    ("random_field", 32, 18, 0, 4, 7),
    ("bvh_stress_500", 24, 14, 0, 2, 7),
]


def rays_for(name, n, rng):
    if name.startswith("cornell"):
        o = rng.uniform(1, 554, (n, 3)).astype(np.float32)
    elif name.startswith("bvh_stress"):
        o = rng.uniform(0, 100, (n, 3)).astype(np.float32)
        o[:, 1] = rng.uniform(0.1, 12, n)
    else:
        o = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
        o[:, 1] = rng.uniform(0.05, 4, n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[::7, 1] = 0.0          # parallel to the XZ rects: t = +-inf
    d[1::11, 0] = 0.0
    d[2::13] *= 1e-3         # short directions (t is in units of |d|)
    d[3::17] *= 1e3
    return o, d


def main():
    assert ref.available(), "build oracle/_ref first"
    out = {}
    for name, w, h, s0, spp, seed in SAMPLE_CASES:
        r = ref.RefScene(name, w, h).render(s0, spp, seed=seed, threads=8, per_sample=True)
        key = f"samples_{name}_{w}x{h}_s{s0}+{spp}_seed{seed}"
        np.savez_compressed(os.path.join(HERE, key + ".npz"), samples=r["samples"], segments=r["segments"].astype(np.uint8)
                            if r["segments"].max() < 256 else r["segments"], sum=r["sum"], rays=np.uint64(r["rays"]),
                            draws=np.uint64(r["draws"]))
        out[key] = dict(scene=name, w=w, h=h, spp_begin=s0, spp=spp, seed=seed, rays=int(r["rays"]))
        print(key, r["rays"])

    rng = np.random.default_rng(2024)
    for name in ("cornell", "cornell_extra", "random_field", "bvh_stress_500"):
        o, d = rays_for(name, 4096, rng)
        hrec = ref.RefScene(name, 32, 32).hit(o, d)
        # tie cases: rays from the box corner along wall intersections etc. are covered by the
        # axis-aligned directions above; add rays that start exactly on a wall plane
        np.savez_compressed(os.path.join(HERE, f"hits_{name}.npz"), origins=o, directions=d, hit=hrec["hit"].astype(np.uint8),
                            t=hrec["t"], p=hrec["p"], n=hrec["n"], mat_kind=hrec["mat_kind"].astype(np.int8))
        print("hits", name, int(hrec["hit"].sum()))

    # Image::write table incl. NaN / negative / > 1 / inf
    x = rng.normal(0.3, 0.6, (4096, 3)).astype(np.float32)
    x[5, 1] = np.nan
    x[7] = np.nan
    x[9, 0] = np.inf
    x[11, 2] = -np.inf
    x[13] = [0.0, 1.0, 0.5]
    x[15] = np.float32(1.0) - np.float32(2 ** -24)
    np.savez_compressed(os.path.join(HERE, "image_write.npz"), rgb=x, rgb8=ref.image_write(x))

    # camera rays of every scene family
    cams = {}
    uv = rng.random((256, 2)).astype(np.float32)
    for name, w, h in (("cornell", 64, 36), ("cornell", 800, 400), ("small", 400, 225), ("bvh_stress_500", 1920, 1080)):
        o, d = ref.RefScene(name, w, h).camera_rays(uv)
        cams[f"{name}_{w}x{h}_o"] = o
        cams[f"{name}_{w}x{h}_d"] = d
    np.savez_compressed(os.path.join(HERE, "camera_rays.npz"), uv=uv, **cams)

    # the reference's BMP writer on a small odd-width image (row padding)
    img = (rng.random((5, 7, 3)) * 255).astype(np.uint8)
    ref.write_bmp("/tmp/_golden.bmp", img)
    np.savez_compressed(os.path.join(HERE, "bmp_7x5.npz"), rgb8=img, bmp_bytes=np.frombuffer(open("/tmp/_golden.bmp", "rb").read(), np.uint8))

    # known answers of the random stream / sincos contract (include/rtx_det.h), from the
    # reference build's own draws: first 8 draws of a few (pixel, sample, seed) keys are implied
    # by the per-sample fixtures above; keep the file list
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)

    # algorithmic flops per ray (op-counting build of the CPU restatement) per scene
    flops = {}
    for name, w, h, spp in (("cornell", 480, 270, 8), ("cornell_classic", 480, 270, 8), ("cornell_extra", 480, 270, 8),
                            ("small", 400, 225, 8), ("random_field", 240, 135, 2)):
        hs = api.HostScene(name, w, h)
        c = restate.count_flops(hs, w, h, 0, spp, seed=0, threads=8)
        flops[name] = dict(flops_per_ray=c["flops"] / c["rays"], rays=c["rays"], flops=c["flops"], w=w, h=h, spp=spp,
                           note="linear scan, as the reference performs it")
        print("flops/ray", name, flops[name]["flops_per_ray"])
    with open(os.path.join(HERE, "flops_per_ray.json"), "w") as f:
        json.dump(flops, f, indent=1, sort_keys=True)

    # high-spp ground truth for the fast-mode statistical test
    w, h, spp = 160, 90, 4096
    hs = api.HostScene("cornell", w, h)
    gt = restate.render(hs, w, h, 1 << 20, spp, seed=99, threads=8)["sum"] / spp
    np.savez_compressed(os.path.join(HERE, "ground_truth_cornell_160x90.npz"), mean=gt.astype(np.float32),
                        spp=np.int32(spp))
    print("ground truth done", gt.mean())


if __name__ == "__main__":
    main()
