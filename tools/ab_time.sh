#!/bin/bash
# A/B timing of library variants on the GPU box: bash tools/ab_time.sh "<variant names>" <gpu_time args...>
# (variant "base" = raytrace_amd/librtx.so, otherwise raytrace_amd/librtx_<name>.so)
VARS=$1; shift
for v in $VARS; do
  if [ "$v" = base ]; then unset RTX_LIBRTX; else export RTX_LIBRTX=$PWD/raytrace_amd/librtx_$v.so; fi
  echo -n "$v: "; timeout -k 10 120 python tools/gpu_time.py "$@" 2>/dev/null | tail -n 1 || exit 1
done
