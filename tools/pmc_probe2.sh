#!/bin/bash
# Second-level PMC probe (instruction mix, fetch, SMEM / VMEM queue levels) of one tools/gpu_time.py configuration.
set -u
TAG=$1; shift
OUT=gpurun_out/pmc2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_MISC -d $OUT/p1 -o p1 -- python3 tools/gpu_time.py "$@" > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES -d $OUT/p2 -o p2 -- python3 tools/gpu_time.py "$@" > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $OUT/p3 -o p3 -- python3 tools/gpu_time.py "$@" > $OUT/p3.log 2>&1 || exit 1
python3 - <<PY
import sqlite3, glob
c = {}
for db in glob.glob("$OUT/p*/*.db"):
    con = sqlite3.connect(db)
    for k, n, v in con.execute("select counter_name, count(*), sum(value) from counters_collection where kernel_name like '%_mode::%' group by counter_name"):
        c[k] = v / n
for k in sorted(c): print(f"{k:26s} {c[k]:20.0f}")
PY
grep Grays $OUT/p1.log
