#!/usr/bin/env python3
"""profiles/hbm_traffic.json (what bench.py reports as roofline.traffic) from the profile summaries under
profiles/: python tools/make_hbm_traffic.py profiles/r03a_bench_*.txt
Each entry records the summary it came from."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
table = json.load(open(out_path)) if os.path.exists(out_path) else {}
for path in sys.argv[1:]:
    text = open(path).read()
    lines = [json.loads(l) for l in text.splitlines() if l.startswith("{")]
    fetch = re.search(r"^FETCH_SIZE\s+([0-9.]+)", text, re.M)
    write = re.search(r"^WRITE_SIZE\s+([0-9.]+)", text, re.M)
    if not lines or not fetch or not write:
        print("skipped (no bench line or no FETCH_SIZE / WRITE_SIZE):", path)
        continue
    serial = lines[-1]  # the PMC passes run with the --serial run's launch size
    wl = serial["config"]["workload"]
    f, w = float(fetch.group(1)), float(write.group(1))
    table[wl] = {
        "bytes_per_launch": int((f + w) * 1024),
        "fetch_kib": f,
        "write_kib": w,
        "spp_per_launch": serial["config"]["spp_per_step_per_gpu"],
        "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, bench.py --serial), {os.path.relpath(path, ROOT)}; "
                  "reads of the linear-scan scenes are scalar / dword traffic, of the BVH scenes 16-byte per-lane node and leaf "
                  "fetches that miss the L2 (no x2 correction applied)",
    }
    print(wl, table[wl]["bytes_per_launch"])
json.dump(table, open(out_path, "w"), indent=1, sort_keys=True)
