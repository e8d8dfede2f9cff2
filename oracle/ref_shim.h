/*
 * ref_shim.h — forced-include header for building the UNMODIFIED reference
 * tracer (/root/reference/Raytrace_C++/Src/*.cpp) with g++ as the parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, loaded by or
 * called from the renderer; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker.
 *
 * The reference sources rely on MSVC's forced include of Pch.h
 * (Raytrace_C++.vcxproj:110-112) and on transitive Windows headers, so this
 * file (a) pre-includes every system header the sources need BEFORE any macro
 * is defined, (b) renames the reference's own `inline float drand48()`
 * (Util.h:10) away from glibc's drand48, (c) substitutes rand()/RAND_MAX
 * (Util.h:11) by the counter-based stream of include/rtx_det.h, (d) routes the
 * sin/cos/tan calls that decide path geometry (Util.h:91-92,104-105,
 * CheckerTexture.cpp:4, Camera.h:13, quat_aos.h:352-353) and the atan2/asin of
 * the sphere texture coordinates (Util.h:79-80) through the bit-reproducible
 * routines of the same header, then (e) includes Pch.h.
 * No reference source text is copied; the sources are compiled where they lie.
 */
#ifndef ORACLE_REF_SHIM_H
#define ORACLE_REF_SHIM_H

#include <cstdlib>
#include <stdlib.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <assert.h>
#include <limits.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <string>
#include <algorithm>
#include <memory>
#include <iostream>
#include <vector>
#include <thread>
#include <atomic>
#include <map>
#include <random>

#include "rtx_det.h"

/* per-thread stream state: set by the driver before each sample */
extern thread_local RtxRng g_rt_rng;
extern thread_local unsigned long long g_rt_draws;

static inline int rt_rand()
{
    ++g_rt_draws;
    return (int)rtx_rng_next24(&g_rt_rng);
}
static inline float rt_sin(float x) { return rtx_sinf(x); }
static inline float rt_cos(float x) { return rtx_cosf(x); }
static inline float rt_tan(float x) { return rtx_tanf(x); }
static inline float rt_atan2(float y, float x) { return rtx_atan2f(y, x); }
static inline float rt_asin(float x) { return rtx_asinf_with_root(x, sqrtf((1.0f - x) * (1.0f + x))); }

#define drand48 rt_reference_drand48
#define rand rt_rand
#undef RAND_MAX
#define RAND_MAX 16777216
#define sinf rt_sin
#define cosf rt_cos
#define tanf rt_tan
#define sin rt_sin
#define cos rt_cos
#define atan2 rt_atan2
#define asin rt_asin

#include "Pch.h"

#endif
