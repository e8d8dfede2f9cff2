"""Compile the kernel rtx_upload_scene would compile for a scene, offline with hipcc, and keep its
assembly: for reading the generated code and its register / instruction counts (no GPU needed).

usage: python tools/special_isa.py [scene [out_dir]]       (default: cornell /tmp/special_isa)
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, ".")
from raytrace_amd import api  # noqa: E402

FASTFLAGS = ["-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=fast", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
             "-fno-math-errno", "-fno-signed-zeros", "-fno-trapping-math", "-Wno-unused-function"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/special_isa"
    os.makedirs(out, exist_ok=True)
    scan = api.scene_kernel_source(api.HostScene(name, 64, 64))
    src = os.path.join(out, name + ".hip")
    with open(src, "w") as f:
        f.write("#define RTX_DET 0\n#define RTX_TRACE_WAVES 8\n" + scan + "#include \"rtx_core.h\"\n"
                "template __global__ void rtx_fast_mode::trace_kernel<false, false, false, true>(const rtxd::TraceArgs);\n")
    asm = os.path.join(out, name + ".s")
    cmd = ["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-Iraytrace_amd/csrc", "-Iinclude", *FASTFLAGS, src, "-o", asm]
    subprocess.run(cmd, check=True)
    text = open(asm).read()
    body = text[text.index("trace_kernel"):]
    counts = {}
    for line in body.splitlines():
        m = re.match(r"\s+([sv]_[a-z0-9_]+|buffer_[a-z0-9_]+|global_[a-z0-9_]+|ds_[a-z0-9_]+|scratch_[a-z0-9_]+)\b", line)
        if m:
            unit = m.group(1).split("_")[0]
            counts[unit] = counts.get(unit, 0) + 1
    print("static instruction counts:", counts)
    for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size"):
        m = re.search(r"\." + key + r":\s+(\d+)", text)
        if m:
            print(key, m.group(1))
    print(asm)


if __name__ == "__main__":
    main()
