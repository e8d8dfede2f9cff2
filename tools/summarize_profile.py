#!/usr/bin/env python3
"""Turn the rocprofv3 result databases written by tools/profile_bench.sh into one text
summary (per-kernel timing from the --kernel-trace --stats run, per-kernel PMC sums from
the --pmc runs, and the figures derived from them).

    python tools/summarize_profile.py gpurun_out/prof_<tag> > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys


def rows(db, q):
    con = sqlite3.connect(db)
    try:
        return list(con.execute(q))
    finally:
        con.close()


def main():
    root = sys.argv[1]
    print(f"# rocprofv3 summary of {root}")
    stats = glob.glob(os.path.join(root, "stats", "*.db"))
    log = os.path.join(root, "stats.log")
    bench_line = None
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                import json

                bench_line = json.loads(line)
                print("\n## bench.py line of the profiled run (rocprofv3 --kernel-trace --stats)\n" + line.strip())
    def trace_section(db, line, title):
        print(f"\n## kernel trace stats, {title} (all dispatches of the run, warm-up included)")
        print(f"{'kernel':90s} {'calls':>6s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}")
        for name, calls, total, avg, pct in rows(db, "select name,total_calls,total_duration,average,percentage from top_kernels"):
            print(f"{name[:90]:90s} {calls:6d} {total:14.1f} {avg:12.1f} {pct:7.2f}")
        print("\n## per-dispatch durations of the path-trace kernel")
        print(f"{'dispatch':>8s} {'start_us':>12s} {'duration_us':>12s} {'grid':>10s} {'wg':>5s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>6s} {'scratch':>8s}")
        spans = []
        t0 = None
        for r in rows(db, "select dispatch_id, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size "
                          "from kernels where name like '%_mode::%' order by start"):
            t0 = r[1] if t0 is None else t0
            print(f"{r[0]:8d} {(r[1] - t0) / 1e3:12.1f} {(r[2] - r[1]) / 1e3:12.1f} {r[3]:10d} {r[4]:5d} {r[5]:5d} {r[6]:5d} {r[7]:5d} {r[8]:6d} {r[9]:8d}")
            spans.append((r[1], r[2]))
        if line and len(spans) >= line["steps"]:
            timed = spans[-line["steps"]:]
            each = sum(e - s for s, e in timed) / len(timed) / 1e3
            span = (max(e for _, e in timed) - min(s for s, _ in timed)) / len(timed) / 1e3
            roof = line["roofline"]
            print(f"the {len(timed)} timed dispatches (the {line['warmup']} warm-up dispatch(es) before them excluded): average duration "
                  f"{each:.1f} us, start of the first to end of the last / {len(timed)} = {span:.1f} us per launch")
            print(f"bench.py's HIP events in the same run: kernel_ms_avg (first start to last end / launches) "
                  f"{roof['kernel_ms_avg'] * 1e3:.1f} us, average of the launches' own durations "
                  f"{roof.get('kernel_ms_each_avg', roof['kernel_ms_avg']) * 1e3:.1f} us, launch streams {roof.get('launch_streams', 1)}")

    if stats:
        trace_section(stats[0], bench_line, "default run")
    serial = glob.glob(os.path.join(root, "stats_serial", "*.db"))
    serial_log = os.path.join(root, "stats_serial.log")
    if serial and os.path.exists(serial_log):
        import json

        serial_line = None
        for line in open(serial_log):
            if line.startswith("{"):
                serial_line = json.loads(line)
                print("\n## bench.py line of the --serial run (one render stream: the kernel alone)\n" + line.strip())
        trace_section(serial[0], serial_line, "--serial run")
    counters = {}
    meta = None
    for db in sorted(glob.glob(os.path.join(root, "pmc*", "*.db"))):
        for kname, cname, n, total, vg, sg, lds in rows(db, "select kernel_name, counter_name, count(*), sum(value), max(vgpr_count), max(sgpr_count), max(lds_block_size) "
                                                       "from counters_collection where kernel_name like '%_mode::%' group by kernel_name, counter_name"):
            counters[cname] = (total / n, n, kname)
            meta = (kname, vg, sg, lds)
    if counters:
        print(f"\n## PMC counters of {meta[0]} (average per dispatch; separate --pmc runs)")
        print(f"   vgpr_count {meta[1]}  sgpr_count {meta[2]}  lds {meta[3]}")
        for k in sorted(counters):
            print(f"{k:28s} {counters[k][0]:22.1f}   (n={counters[k][1]})")
        c = {k: v[0] for k, v in counters.items()}
        print("\n## derived")
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            print(f"VALU active-lane fraction  = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}")
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over SIMDs; SQ_BUSY_CYCLES per SE
            pass
        if "GRBM_GUI_ACTIVE" in c:
            print(f"GRBM_GUI_ACTIVE / 8 XCDs   = {c['GRBM_GUI_ACTIVE'] / 8:.0f} cycles per dispatch")
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles; 1024 SIMDs
            print(f"VALU issue-slot occupancy  = SQ_INSTS_VALU * 2 / (1024 SIMDs * cycles) = {c['SQ_INSTS_VALU'] * 2 / (1024 * cyc):.3f}")
        if "SQ_WAVE_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            print(f"mean resident waves / SIMD = SQ_WAVE_CYCLES * 4 / (1024 SIMDs * cycles) = {c['SQ_WAVE_CYCLES'] * 4 / (1024 * cyc):.2f}   (SQ_WAVE_CYCLES counts quad-cycles)")
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in c and "SQ_WAVE_CYCLES" in c:
                print(f"{k} / SQ_WAVE_CYCLES = {c[k] / c['SQ_WAVE_CYCLES']:.3f}")
        if "FETCH_SIZE" in c:
            print(f"HBM read  per dispatch     = FETCH_SIZE {c['FETCH_SIZE']:.1f} KiB (x2 gfx950 correction for wide reads not applied: scalar / dword traffic)")
        if "WRITE_SIZE" in c:
            print(f"HBM write per dispatch     = WRITE_SIZE {c['WRITE_SIZE']:.1f} KiB")
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            print(f"L2 hit rate                = TCC_HIT / (TCC_HIT + TCC_MISS) = {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.4f}")
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
            print(f"vector L1 hit rate         = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES = {1 - c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")


if __name__ == "__main__":
    main()
