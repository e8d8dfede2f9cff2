#!/bin/bash
# Profiles of every bench workload on the GPU box, reduced to text summaries there (the rocprofv3
# databases are too large to travel back).  usage (via gpurun): bash tools/profile_all.sh <round tag>
# Writes gpurun_out/profiles_<tag>/<tag>_bench_<workload>.txt; copy them into profiles/ and run
# tools/make_hbm_traffic.py on them.
set -u
TAG=${1:-run}
DST=gpurun_out/profiles_$TAG
mkdir -p $DST
for pair in cornell:cornell_1080p:12:3 field:random_field_1080p:4:1 stress:bvh_stress_1080p:4:1 4k:cornell_extra_4k:8:2 small:small_400x225:12:3; do
  IFS=: read short wl steps warm <<< "$pair"
  bash tools/profile_workload.sh ${TAG}_$short $wl $steps $warm > $DST/$short.log 2>&1 || exit 1
  python3 tools/summarize_profile.py gpurun_out/prof_${TAG}_$short > $DST/${TAG}_bench_${wl}.txt || exit 1
  rm -rf gpurun_out/prof_${TAG}_$short
done
ls -la $DST
