#!/bin/bash
# A/B/... of several builds of librtx.so on the GPU box, alternating runs (the clocks of a box drift):
#   tools/ab.sh "scene w h spp mode depth accel" rounds libA.so libB.so ...
# prints one gpu_time.py line per run, prefixed with the library.
set -e
cfg="$1"; rounds="$2"; shift 2
for r in $(seq "$rounds"); do
  for lib in "$@"; do
    echo -n "[$(basename "$lib")] "
    RTX_LIBRTX="$PWD/$lib" timeout -k 10 120 python tools/gpu_time.py $cfg 3 2>&1 | grep -v amdgpu.ids
  done
done
