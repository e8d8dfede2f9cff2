"""Per-basic-block instruction mix of one kernel in an AMDGPU .s file.
usage: python tools/isa_blocks.py file.s kernel_substring"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l and not l.startswith("\t"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks = []
cur = {"label": "entry", "v": 0, "s": 0, "smem": 0, "vmem": 0, "lds": 0, "br": [], "first": start}
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = {"label": l.split(":")[0], "v": 0, "s": 0, "smem": 0, "vmem": 0, "lds": 0, "br": [], "first": i}
        continue
    op = l.split()[0]
    if op.startswith("v_"):
        cur["v"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer_load"):
        cur["smem"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur["br"].append(l.split()[1])
        cur["s"] += 1
    elif op.startswith("s_"):
        cur["s"] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        cur["vmem"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
blocks.append(cur)
tot = {k: sum(b[k] for b in blocks) for k in ("v", "s", "smem", "vmem", "lds")}
print("total", tot, "blocks", len(blocks))
for b in blocks:
    print(f'{b["label"]:>12} line {b["first"]:6d}  v={b["v"]:4d} s={b["s"]:4d} smem={b["smem"]:3d} vmem={b["vmem"]:3d} lds={b["lds"]:3d}  -> {",".join(b["br"])}')
