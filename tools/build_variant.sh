#!/bin/bash
# Build a variant of librtx.so that differs only in the fast-mode kernels:
#   bash tools/build_variant.sh <name> <-DMACRO[=value] ...>   ->  raytrace_amd/librtx_<name>.so
# The defines go to the fast-mode translation unit and, as #define lines, to the kernels the variant compiles per
# scene at run time; other compiler flags reach the translation unit only.  The other objects come from
# raytrace_amd/csrc/build, so run `make -C raytrace_amd/csrc` first.
# Use with RTX_LIBRTX=raytrace_amd/librtx_<name>.so for A/B timing on the GPU box (tools/ab.sh).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../raytrace_amd/csrc"
mkdir -p build_var
DEFS=""
for flag in "$@"; do
  case "$flag" in
    -D*=*) m=${flag#-D}; DEFS="$DEFS#define ${m%%=*} ${m#*=}\\n" ;;
    -D*) DEFS="$DEFS#define ${flag#-D} 1\\n" ;;
  esac
done
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $COMMON \
  -DRTX_TRACE_WAVES=8 -fno-slp-vectorize -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-math-errno \
  -fno-signed-zeros -fno-trapping-math "$@" -c rtx_kernels_fast.hip -o build_var/fast_$NAME.o
/opt/rocm/bin/hipcc $COMMON -x hip "-DRTX_VARIANT_DEFINES=\"$DEFS\"" -c rtx_special.cpp -o build_var/special_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librtx_$NAME.so build/rtx_kernels_det.o build_var/fast_$NAME.o \
  build/rtx_resolve.o build/rtx_api.o build/rtx_bvh.o build_var/special_$NAME.o build/rtx_sources.o -ldl
