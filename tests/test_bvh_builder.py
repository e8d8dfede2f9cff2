"""Host logic of the two tree builders (raytrace_amd/csrc/rtx_bvh.cpp), without a GPU: the checker
tests/bvh_check.cpp is compiled with g++ against the builder's source and run on random scenes
(bare spheres and rects, rects under Rotate + Translate chains, one oversized item).

Scene tree: every group in exactly one leaf or kept outside the tree, depth within the traversal
stack, a ray that meets an item's exact bounds reaches its leaf through the decoded 16-bit boxes.
Light tree: depth-first order, skip links past each subtree, every entry in one leaf, the
stackless walk finds every light whose bounds a ray meets.

The checker is built with AddressSanitizer and UndefinedBehaviorSanitizer (host code only)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvh") / "bvh_check")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "raytrace_amd", "csrc"),
           os.path.join(ROOT, "tests", "bvh_check.cpp"), os.path.join(ROOT, "raytrace_amd", "csrc", "rtx_bvh.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (3, 7), (40, 3), (500, 4), (5000, 5), (20000, 6)])
def test_tree_builders_keep_their_invariants(checker, n, seed):
    # (a preloaded library may sit in front of the sanitizer runtime: do not insist on the link order)
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([checker, str(n), str(seed)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.splitlines()[-1].startswith("ok"), r.stdout + r.stderr
    assert "estimated visits=" in r.stdout
