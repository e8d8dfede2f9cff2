"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU): the scene flattener and
the host image pipeline (raytrace_amd/host/raytrace.cpp, scene_factory.cpp), driven by
tests/flatten_check.cpp over every named scene.  The tree builders have their own sanitizer run
in tests/test_bvh_builder.py.  Device code cannot be sanitized on this pool; these are the parts of
the library that run on the host for every scene upload.  The JPEG decoder behind ImageTexture
reads files from outside: it is run on damaged files as well.  The host side of the run-time
compilation (raytrace_amd/csrc/rtx_special.cpp: source generation, the compiler's entry points taken
from the process, the cache) runs under the sanitizers through tests/special_check.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flattener_and_image_pipeline_are_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "flatten_check")
    host = os.path.join(ROOT, "raytrace_amd", "host")
    # the render entry points of the host library call rtx_*: link the real library (its HIP
    # runtime is loaded but never initialised: nothing here renders)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include"), "-I" + host,
           os.path.join(ROOT, "tests", "flatten_check.cpp"), os.path.join(host, "raytrace.cpp"),
           os.path.join(host, "scene_factory.cpp"), os.path.join(host, "jpeg_baseline.cpp"),
           "-L" + os.path.join(ROOT, "raytrace_amd"), "-lrtx",
           "-Wl,-rpath," + os.path.join(ROOT, "raytrace_amd"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0:protect_shadow_gap=0",
               UBSAN_OPTIONS="print_stacktrace=1",
               RTX_TEXTURE=os.path.join(ROOT, "tests", "golden", "texture_64x32.bmp"))
    r = subprocess.run([exe, str(tmp_path / "o.bmp")], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]
    assert os.path.getsize(str(tmp_path / "o.bmp")) == 54 + 5 * 24  # 7 pixels * 3 bytes padded to 24 per row


def test_jpeg_decoder_survives_damaged_files_under_asan_ubsan(tmp_path):
    """tests/jpeg_check.cpp: every fixture file intact, then 3 000 damaged copies of each."""
    import numpy as np

    exe = str(tmp_path / "jpeg_check")
    host = os.path.join(ROOT, "raytrace_amd", "host")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + host, os.path.join(ROOT, "tests", "jpeg_check.cpp"), os.path.join(host, "jpeg_baseline.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    g = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
    files = []
    for key in g.files:
        if key.endswith("_file"):
            files.append(str(tmp_path / (key[:-5] + ".jpg")))
            with open(files[-1], "wb") as f:
                f.write(g[key].tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]
    decoded, refused = (int(x) for x in r.stdout.split()[1:3])
    assert decoded > 1000 and refused > 1000  # both outcomes were exercised


def test_run_time_compilation_host_code_is_clean_under_asan_ubsan(tmp_path):
    """tests/special_check.cpp (a scan that does not compile, the same again, a good one) with the host code of
    rtx_special.cpp instrumented; the compiler library itself is not."""
    csrc = os.path.join(ROOT, "raytrace_amd", "csrc")
    exe = str(tmp_path / "special_check_asan")
    sources = str(tmp_path / "rtx_sources.o")
    r = subprocess.run(["g++", "-O1", "-c", "rtx_sources.cpp", "-o", sources], capture_output=True, text=True, cwd=csrc)
    assert r.returncode == 0, r.stderr[-2000:]
    # (hipcc: the sanitizer flags go to the host compilation only)
    cmd = ["hipcc", "-std=c++17", "-O1", "-g", "-x", "hip", "--offload-arch=gfx950", "-Xarch_host", "-fsanitize=address",
           "-Xarch_host", "-fsanitize=undefined", "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "special_check.cpp"), os.path.join(csrc, "rtx_special.cpp"), "-x", "none", sources,
           "-ldl", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr[-3000:]
