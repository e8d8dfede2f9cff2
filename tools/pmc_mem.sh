#!/bin/bash
# Memory-hierarchy PMC probe of one tools/gpu_time.py configuration (GPU box): vector L1 (TCP) and
# L2 (TCC) request / hit / miss counts and the texture-addresser busy share, one pass per block limit.
#   bash tools/pmc_mem.sh <tag> <gpu_time args...>
set -u
TAG=$1; shift
OUT=gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/p1 -o p1 -- python3 tools/gpu_time.py "$@" > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/p2 -o p2 -- python3 tools/gpu_time.py "$@" > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d $OUT/p3 -o p3 -- python3 tools/gpu_time.py "$@" > $OUT/p3.log 2>&1 || exit 1
python3 - <<PY
import sqlite3, glob
c = {}
for db in glob.glob("$OUT/p*/*.db"):
    con = sqlite3.connect(db)
    for k, n, v in con.execute("select counter_name, count(*), sum(value) from counters_collection where kernel_name like '%_mode::%' group by counter_name"):
        c[k] = v / n
for k in sorted(c): print(f"{k:34s} {c[k]:20.0f}")
PY
grep -h Grays $OUT/p1.log
