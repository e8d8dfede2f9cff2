#!/bin/bash
# Kernel-trace statistics of bench.py for one named workload (GPU box).
# usage (via gpurun): bash tools/profile_workload.sh <tag> <workload> [steps]
set -u
TAG=$1; WL=$2; STEPS=${3:-4}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $OUT/pmc1 -o pmc1 -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --serial > $OUT/pmc1.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --serial > $OUT/pmc2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --serial > $OUT/pmc3.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --serial > $OUT/pmc4.log 2>&1 || exit 1
grep '^{' $OUT/stats.log | cut -c1-300
