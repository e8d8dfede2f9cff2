// Fast-mode kernels: same algorithm and random stream as the deterministic build,
// compiled with floating-point contraction on and hardware rcp / rsq / sin / cos.
#define RTX_DET 0
#include "rtx_core.h"
#include "rtx_launch.h"
#include "rtx_kernels.inc"
