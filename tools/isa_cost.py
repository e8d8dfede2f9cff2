"""Estimated VALU issue cycles per basic block of one kernel in an AMDGPU .s file, using the
per-instruction costs measured by tools/probe/valu_cost.hip on gfx950 (cycles per wave64).
usage: python tools/isa_cost.py file.s kernel_substring"""
import re
import sys

FULL, HALF, QUARTER = 2.5, 4.2, 8.2
HALF_OPS = ("v_cmp", "v_min", "v_max", "v_med3", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i", "v_cvt", "v_and_or",
            "v_or3", "v_xor3", "v_add3", "v_lshl_add", "v_lshl_or", "v_add_lshl", "v_bfe", "v_bfi", "v_alignbit", "v_perm",
            "v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "v_pk_", "v_ldexp", "v_frexp", "v_fract",
            "v_floor", "v_trunc", "v_rndne", "v_ceil", "v_div_")
QUARTER_OPS = ("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")


def cost(line):
    op = line.split()[0]
    if any(op.startswith(q) for q in QUARTER_OPS):
        return QUARTER, "T"
    if any(op.startswith(h) for h in HALF_OPS):
        return HALF, "H"
    args = line[len(op):].split(";")[0]
    srcs = args.split(",")[1:]  # skip destination
    if op.startswith("v_cndmask") and "_e64" not in op:
        srcs = srcs[:2]  # the implicit vcc mask is free
    if any(re.search(r"\b(s\d+|s\[\d+:\d+\]|vcc|exec)\b", a) for a in srcs):
        return HALF, "S"
    return FULL, "F"


path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], None


def new(label, i):
    return {"label": label, "first": i, "cyc": 0.0, "n": 0, "kinds": {"F": 0, "H": 0, "S": 0, "T": 0}, "salu": 0, "mem": 0}


cur = new("entry", start)
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = new(m.group(1), i)
        continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    if op.startswith("v_"):
        c, k = cost(l)
        cur["cyc"] += c
        cur["n"] += 1
        cur["kinds"][k] += 1
    elif op.startswith("s_load") or op.startswith("global_") or op.startswith("ds_") or op.startswith("buffer_"):
        cur["mem"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
blocks.append(cur)
print(f"total VALU {sum(b['n'] for b in blocks)}  est. cycles if every block ran once: {sum(b['cyc'] for b in blocks):.0f}")
for b in blocks:
    k = b["kinds"]
    print(f'{b["label"]:>12} line {b["first"]:6d}  valu={b["n"]:4d} cyc={b["cyc"]:7.1f}  full={k["F"]:3d} half={k["H"]:3d} sgpr={k["S"]:3d} trans={k["T"]:2d}  salu={b["salu"]:3d} mem={b["mem"]:2d}')
