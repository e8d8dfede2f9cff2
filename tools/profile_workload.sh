#!/bin/bash
# rocprofv3 evidence for bench.py on one named workload (GPU box).
# usage (via gpurun): bash tools/profile_workload.sh <tag> <workload> [steps] [warmup]
# Writes gpurun_out/prof_<tag>/...: the kernel trace of the default run (consecutive launches overlap on two
# streams), the kernel trace of a --serial run (the kernel alone), and the PMC passes, each a run of its own
# with --serial (<= 8 SQ counters per pass; nothing else on the GPU while the trace kernel runs).
set -u
TAG=$1; WL=$2; STEPS=${3:-12}; WARM=${4:-3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --workload $WL --no-cpu-baseline"
SER="$ARGS --steps 4 --warmup 1 --serial"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ARGS --steps $STEPS --warmup $WARM > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_serial -o stats -- python3 $SER > $OUT/stats_serial.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $OUT/pmc1 -o pmc1 -- python3 $SER > $OUT/pmc1.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python3 $SER > $OUT/pmc2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python3 $SER > $OUT/pmc3.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python3 $SER > $OUT/pmc4.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr -d $OUT/pmc5 -o pmc5 -- python3 $SER > $OUT/pmc5.log 2>&1 || exit 1
grep '^{' $OUT/stats.log | cut -c1-300
