/*
 * rtx_det.h — the deterministic-mode contract of the renderer.
 *
 * Deterministic mode replaces the reference tracer's host RNG
 * (Raytrace_C++/Src/Util.h:10-12, `rand()/RAND_MAX`) by a counter-based stream
 * keyed by (pixel, absolute sample index, seed), and routes the few
 * transcendental calls that decide path geometry (Util.h:91-92, 104-105;
 * CheckerTexture.cpp:4; Camera.h:13; quat_aos.h:352-353) through the fp32
 * routines below.  Every routine here uses only IEEE-754 fp32 add / sub / mul /
 * div, integer ops and float<->int conversions, so a translation unit compiled
 * WITHOUT floating-point contraction (-ffp-contract=off) produces bit-identical
 * results on an x86-64 host and on a gfx950 device.  The GPU kernels, the host
 * scene API, the CPU restatement and the build of the reference tracer used as
 * oracle all include this one file; nothing else is shared between them.
 *
 * Plain C99 / C++ / HIP.  No dependencies.
 */
#ifndef RTX_DET_H
#define RTX_DET_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RTX_HD __host__ __device__ __forceinline__
#else
#define RTX_HD static inline
#endif

/* ------------------------------------------------------------------------- */
/* Counter-based random stream                                                 */
/* ------------------------------------------------------------------------- */

/* One 32-bit avalanche step (PCG-RXS-M-XS output permutation over one LCG step). */
RTX_HD uint32_t rtx_hash32(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

/* Per-sample generator: a 32-bit LCG whose odd increment selects the stream. */
typedef struct RtxRng {
    uint32_t state;
    uint32_t inc;
} RtxRng;

/*
 * Key the stream for one (pixel, sample).  `pixel` is j*width+i with j counted
 * from the BOTTOM image row (the reference's loop variable, Scene.cpp:140-146);
 * `sample` is the absolute sample index, so splitting the sample range over
 * several GPUs does not change any stream.
 */
RTX_HD void rtx_rng_init(RtxRng* r, uint32_t pixel, uint32_t sample, uint64_t seed)
{
    uint32_t k = rtx_hash32((uint32_t)seed ^ 0x9e3779b9u);
    k = rtx_hash32(k ^ (uint32_t)(seed >> 32));
    uint32_t a = rtx_hash32(pixel ^ k);
    uint32_t b = rtx_hash32(sample + 0x85ebca6bu * (k | 1u));
    r->state = rtx_hash32(a ^ (b * 0x9e3779b9u + 0x7f4a7c15u));
    r->inc = (rtx_hash32(a + (b ^ 0x27d4eb2fu)) << 1) | 1u;
}

/* Next 24-bit integer of the stream (what the oracle build hands out as rand()). */
RTX_HD uint32_t rtx_rng_next24(RtxRng* r)
{
    uint32_t s = r->state * 747796405u + r->inc;
    r->state = s;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return ((w >> 22) ^ w) >> 8;
}

/*
 * Next uniform float in [0,1): k * 2^-24, exactly the value the reference's
 * drand48() (Util.h:10-12) returns when rand() yields k and RAND_MAX is 2^24.
 */
RTX_HD float rtx_rng_nextf(RtxRng* r)
{
    return (float)rtx_rng_next24(r) * 5.9604644775390625e-8f;
}

/* ------------------------------------------------------------------------- */
/* fp32 sine / cosine / tangent, bit-reproducible across host and device       */
/* ------------------------------------------------------------------------- */

/*
 * Cody-Waite reduction by pi/2 in three pieces followed by the classic
 * single-precision minimax polynomials on [-pi/4, pi/4].  Absolute error is
 * about 1e-7 for |x| < 1e3 and grows linearly beyond (the three-piece constant
 * carries ~33 bits); arguments here are 2*pi*xi (samplers) and freq*coordinate
 * (checker texture).  |x| must stay below 2^22.
 */
RTX_HD void rtx_sincosf(float x, float* sn, float* cs)
{
    float y = x * 0.636619772367581343f; /* x * 2/pi */
    /* round to nearest integer without libm: truncation of y +- 0.5 */
    int k = (int)(y + (y < 0.0f ? -0.5f : 0.5f));
    float kf = (float)k;
    float r = x - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    ps = ps * z * r + r;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    pc = pc * z * z - 0.5f * z + 1.0f;
    switch (k & 3) {
    case 0: *sn = ps; *cs = pc; break;
    case 1: *sn = pc; *cs = -ps; break;
    case 2: *sn = -ps; *cs = -pc; break;
    default: *sn = -pc; *cs = ps; break;
    }
}

RTX_HD float rtx_sinf(float x)
{
    float s, c;
    rtx_sincosf(x, &s, &c);
    return s;
}

RTX_HD float rtx_cosf(float x)
{
    float s, c;
    rtx_sincosf(x, &s, &c);
    return c;
}

RTX_HD float rtx_tanf(float x)
{
    float s, c;
    rtx_sincosf(x, &s, &c);
    return s / c;
}

/*
 * fp32 atan2 / asin for the sphere texture coordinates (get_sphere_uv, Util.h:78-83), again
 * add / sub / mul / div only.  atan on [0,1] is an odd minimax polynomial (error < 2e-7 rad
 * before rounding), extended by atan(x) = pi/2 - atan(1/x) and the usual quadrant rules; the
 * square root in rtx_asinf must be IEEE-correct on both sides (it is: sqrtf / v_sqrt with
 * correct rounding requested).
 */
RTX_HD float rtx_atan_unit(float x) /* 0 <= x <= 1 */
{
    float z = x * x;
    float p = 2.78569828e-3f;
    p = p * z - 1.58660226e-2f;
    p = p * z + 4.24722321e-2f;
    p = p * z - 7.49753043e-2f;
    p = p * z + 1.06448799e-1f;
    p = p * z - 1.42070308e-1f;
    p = p * z + 1.99934542e-1f;
    p = p * z - 3.33331466e-1f;
    return x + x * z * p;
}

RTX_HD float rtx_atan2f(float y, float x)
{
    float ax = x < 0.0f ? -x : x;
    float ay = y < 0.0f ? -y : y;
    float hi = ax > ay ? ax : ay;
    float lo = ax > ay ? ay : ax;
    float r = hi == 0.0f ? 0.0f : rtx_atan_unit(lo / hi);
    if (ay > ax) r = 1.57079632679f - r;
    if (x < 0.0f) r = 3.14159265359f - r;
    return y < 0.0f ? -r : r;
}

/* `root` = sqrtf((1 - x) * (1 + x)) supplied by the caller (the square root is not in this header). */
RTX_HD float rtx_asinf_with_root(float x, float root) { return rtx_atan2f(x, root); }

#endif /* RTX_DET_H */
