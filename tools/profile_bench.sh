#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box.  Usage (via gpurun):
#   bash tools/profile_bench.sh <tag>
# Writes gpurun_out/prof_<tag>/... ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ARGS > $OUT/stats.log 2>&1 || exit 1
# PMC passes, one run each (<= 8 SQ counters per pass); --serial: nothing else runs on the GPU
# while the trace kernel does, so the per-dispatch counters are the kernel's own
ARGS="$ARGS --serial"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $OUT/pmc1 -o pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -50
