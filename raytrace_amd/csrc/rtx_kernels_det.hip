// Deterministic-mode kernels: compiled with -ffp-contract=off and IEEE-correct
// fp32 divide / sqrt so that every operation rounds as it does in the reference
// tracer built by g++ for baseline x86-64.
#define RTX_DET 1
#include "rtx_core.h"
#include "rtx_launch.h"
#include "rtx_kernels.inc"
