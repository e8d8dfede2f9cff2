#!/bin/bash
# Build a variant of librtx.so that differs only in the fast-mode kernel translation unit:
#   bash tools/build_variant.sh <name> <extra compiler flags...>   ->  raytrace_amd/librtx_<name>.so
# (the other objects come from raytrace_amd/csrc/build, so run `make -C raytrace_amd/csrc` first).
# Use with RTX_LIBRTX=raytrace_amd/librtx_<name>.so for A/B timing on the GPU box.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../raytrace_amd/csrc"
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function \
  -DRTX_TRACE_WAVES=8 -fno-slp-vectorize -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-math-errno \
  -fno-signed-zeros -fno-trapping-math "$@" -c rtx_kernels_fast.hip -o build_var/fast_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librtx_$NAME.so build/rtx_kernels_det.o build_var/fast_$NAME.o \
  build/rtx_resolve.o build/rtx_api.o build/rtx_bvh.o
