#!/bin/bash
# PMC probe of one tools/gpu_time.py configuration.  usage: bash tools/pmc_probe.sh <tag> <gpu_time args...>
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/p1 -o p1 -- python3 tools/gpu_time.py "$@" > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -d $OUT/p2 -o p2 -- python3 tools/gpu_time.py "$@" > $OUT/p2.log 2>&1 || exit 1
python3 - <<PY
import sqlite3, glob
c = {}
for db in glob.glob("$OUT/p*/*.db"):
    con = sqlite3.connect(db)
    for k, n, v in con.execute("select counter_name, count(*), sum(value) from counters_collection where kernel_name like '%_mode::%' group by counter_name"):
        c[k] = v / n
for k in sorted(c): print(f"{k:26s} {c[k]:20.0f}")
cyc = c["GRBM_GUI_ACTIVE"] / 8
print("cycles/dispatch", cyc)
print("VALU util lanes", c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]))
print("VALU slot occ  ", c["SQ_INSTS_VALU"] * 2 / (1024 * cyc))
print("SALU per VALU  ", c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"])
print("waves/SIMD     ", c["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc))
print("instr per wave ", (c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_SMEM"] + c["SQ_INSTS_VMEM_RD"]) / c["SQ_WAVES"])
print("wave life cyc  ", c["SQ_WAVE_CYCLES"] * 4 / c["SQ_WAVES"])
PY
grep Grays $OUT/p1.log
