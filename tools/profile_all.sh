#!/bin/bash
# Profiles of every bench workload on the GPU box, reduced to text summaries there (the rocprofv3
# databases are too large to travel back).  usage (via gpurun): bash tools/profile_all.sh <round tag>
# Writes gpurun_out/profiles_<tag>/<tag>_bench_<workload>.txt; copy them into profiles/.
set -u
TAG=${1:-run}
DST=gpurun_out/profiles_$TAG
mkdir -p $DST
bash tools/profile_bench.sh ${TAG}_cornell > $DST/cornell.log 2>&1 || exit 1
python3 tools/summarize_profile.py gpurun_out/prof_${TAG}_cornell > $DST/${TAG}_bench_cornell_1080p.txt || exit 1
rm -rf gpurun_out/prof_${TAG}_cornell
for pair in field:random_field_1080p:2 stress:bvh_stress_1080p:2 4k:cornell_extra_4k:4 small:small_400x225:8; do
  IFS=: read short wl steps <<< "$pair"
  bash tools/profile_workload.sh ${TAG}_$short $wl $steps > $DST/$short.log 2>&1 || exit 1
  python3 tools/summarize_profile.py gpurun_out/prof_${TAG}_$short > $DST/${TAG}_bench_${wl}.txt || exit 1
  rm -rf gpurun_out/prof_${TAG}_$short
done
ls -la $DST
