#!/bin/bash
# LDS-side PMC probe of one tools/gpu_time.py configuration.
set -u
TAG=$1; shift
OUT=gpurun_out/pmc3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAVE_CYCLES -d $OUT/p1 -o p1 -- python3 tools/gpu_time.py "$@" > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d $OUT/p2 -o p2 -- python3 tools/gpu_time.py "$@" > $OUT/p2.log 2>&1 || exit 1
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
