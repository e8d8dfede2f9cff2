// rtx_core.h — device code of the path tracer: nearest-hit query, material
// scattering, light / cosine importance sampling and the iterative integrator.
//
// Included by exactly two translation units which differ only in RTX_DET and in
// their floating-point compile flags:
//   RTX_DET = 1  (rtx_kernels_det.hip, -ffp-contract=off, IEEE divide / sqrt):
//       every fp32 operation is performed in the order the reference tracer
//       performs it, so hit / miss decisions, sampled directions and hit points
//       are bit-identical to the reference fed the same random stream.
//   RTX_DET = 0  (rtx_kernels_fast.hip, contraction on, v_rcp / v_rsq / v_sin):
//       the same estimator, with arithmetic free to fuse and approximate, a lighter
//       random stream, the scan re-ordered by primitive kind, boxes by slab tests and
//       the densities algebraically folded; statistically equal, not bit-equal.
//
// Reference citations are relative to Raytrace_C++/Src.
#pragma once

#ifndef __HIPCC_RTC__  // (the run-time compiler brings the HIP device headers by itself)
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "rtx.h"
#include "rtx_det.h"
#include "rtx_device.h"

#ifndef RTX_DET
#error "define RTX_DET to 0 or 1"
#endif

#if RTX_DET
#define RTX_NS rtx_det_mode
#else
#define RTX_NS rtx_fast_mode
#endif

#define RTX_DEV __device__ __forceinline__

// threads per block of the trace kernels: 256 (2x2 waves, each with its own pixel tile) or 64
#ifndef RTX_BLOCK
#define RTX_BLOCK 256
#endif
// minimum resident waves per SIMD the trace kernel is compiled for (bounds its VGPR budget)
#ifndef RTX_TRACE_WAVES
#define RTX_TRACE_WAVES 8
#endif
#ifndef RTX_TRACE_SGPRS
#define RTX_TRACE_SGPRS 96
#endif

namespace RTX_NS {

using namespace rtxd;

constexpr float kPi = 3.14159265359f;   // Util.h:2
constexpr float kPi2 = 6.28318530718f;  // Util.h:3
constexpr float kTMin = 0.001f;         // Scene.cpp:105, Sphere.cpp:40, Rect.cpp:39
constexpr float kFltMax = 3.402823466e+38f;
constexpr uint32_t kNone = 0xffffffffu;

// A product rounded to fp32 on its own, whatever the contraction mode: the compiler cannot see
// through the empty statement, so it cannot fuse the multiplication into a later add.
RTX_DEV float rounded_product(float x, float y) {
    float p = x * y;
    asm("" : "+v"(p));
    return p;
}

// Discriminant of a t^2 + 2 h t + c = 0 for a sphere.  The reference computes it from two
// ROUNDED products (Sphere.cpp:9-12, b^2 - 4ac), and its pictures depend on that: a ray inside a
// glass sphere is reflected dozens of times at the same angle of incidence, the rounding of each
// hit moves that angle a little, and how many such rays cross the critical angle - leave the
// sphere - before the depth limit ends them is decided by the size of those errors.  A fused
// h * h - a * c (one rounding) keeps the rays on their orbits better and lets 15 % fewer of them
// run into the depth limit, which makes the glass sphere of the Cornell box 1 - 2 % brighter than
// the reference renders it (measured: 16 x 9 cells of the 1080p frame, eight sample ranges per
// mode, t-tests).  So the fast mode rounds here as the reference does.
RTX_DEV float sphere_discriminant(float h, float a, float c) { return rounded_product(h, h) - rounded_product(a, c); }

// ---------------------------------------------------------------------------
// scalar helpers whose implementation differs between the two modes
// ---------------------------------------------------------------------------
#if RTX_DET
RTX_DEV float fdiv(float a, float b) { return a / b; }
RTX_DEV float frcp(float a) { return 1.0f / a; }
RTX_DEV float fsqrt(float a) { return __builtin_sqrtf(a); }
RTX_DEV float finvsqrt(float a) { return 1.0f / __builtin_sqrtf(a); }
// sin / cos of 2*pi*r (Util.h:90-92, 103-105: phi = PI2 * r1)
RTX_DEV void sincos_2pi(float r, float& s, float& c) { rtx_sincosf(kPi2 * r, &s, &c); }
RTX_DEV float fsin(float x) { return rtx_sinf(x); }
#else
RTX_DEV float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
RTX_DEV float frcp(float a) { return __builtin_amdgcn_rcpf(a); }
RTX_DEV float fsqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
RTX_DEV float finvsqrt(float a) { return __builtin_amdgcn_rsqf(a); }
// v_sin_f32 / v_cos_f32 take their argument in revolutions
RTX_DEV void sincos_2pi(float r, float& s, float& c) {
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
}
RTX_DEV float fsin(float x) {
    float rev = x * 0.15915494309189535f;
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev));
}
#endif

// ---------------------------------------------------------------------------
// Vector3 arithmetic in the operation order of vec_aos.h:263-493
// ---------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
RTX_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RTX_DEV V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
RTX_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RTX_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RTX_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RTX_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RTX_DEV V3 vmul(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }  // mulPerElem
RTX_DEV V3 vdivs(V3 a, float s) { return V3{fdiv(a.x, s), fdiv(a.y, s), fdiv(a.z, s)}; }
RTX_DEV float dot(V3 a, V3 b) {  // vec_aos.h:451-458
    float r = a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
RTX_DEV float length_sqr(V3 a) { return dot(a, a); }
RTX_DEV float length(V3 a) { return fsqrt(dot(a, a)); }
RTX_DEV V3 normalize(V3 a) {  // vec_aos.h:474-485: v * (1 / sqrt(len2))
    float inv = finvsqrt(dot(a, a));
    return a * inv;
}
RTX_DEV V3 cross(V3 a, V3 b) {  // vec_aos.h:487-494
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// rotate(quat, vec), quat_aos.h:400-412
RTX_DEV V3 qrotate(float qx, float qy, float qz, float qw, V3 v) {
    float tx = ((qw * v.x) + (qy * v.z)) - (qz * v.y);
    float ty = ((qw * v.y) + (qz * v.x)) - (qx * v.z);
    float tz = ((qw * v.z) + (qx * v.y)) - (qy * v.x);
    float tw = ((qx * v.x) + (qy * v.y)) + (qz * v.z);
    return V3{(((tw * qx) + (tx * qw)) - (ty * qz)) + (tz * qy),
              (((tw * qy) + (ty * qw)) - (tz * qx)) + (tx * qz),
              (((tw * qz) + (tz * qw)) - (tx * qy)) + (ty * qx)};
}

// The lanes of the wave for which `pred` holds.  (HIP's __ballot takes an int: the predicate is
// first turned into 0 / 1 per lane and then compared with 0 again - two vector instructions per
// ballot; the builtin reads the condition mask itself.)
RTX_DEV unsigned long long rtx_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// Record i of a table, addressed per lane with a 32-bit byte offset: the load then takes the table's
// base from a scalar register pair and the offset from one vector register, and the lane does no
// 64-bit address arithmetic (a 64-bit shift and a 64-bit add per load otherwise).  The upload keeps
// every table addressed this way below 4 GiB.
template <typename T>
RTX_DEV const T& at32(const T* base, uint32_t i) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (uint32_t)(i * (uint32_t)sizeof(T)));
}

// Wave-uniform record fetch: the index is the same in every lane, so tell the
// compiler (scalar loads, SGPR operands).
template <typename T>
RTX_DEV T load_uniform(const T* base, uint32_t i) {
    return base[(uint32_t)__builtin_amdgcn_readfirstlane(i)];
}

// ---------------------------------------------------------------------------
// random stream
// ---------------------------------------------------------------------------
#if RTX_DET
// the stream every party of a parity test shares (include/rtx_det.h)
RTX_DEV void rng_init(RtxRng* r, uint32_t pixel, uint32_t sample, uint64_t seed) { rtx_rng_init(r, pixel, sample, seed); }
RTX_DEV float rng_nextf(RtxRng* r) { return rtx_rng_nextf(r); }
#else
// Fast mode only has to be statistically equal, and the shared stream is a sizeable part of
// the kernel's issue cycles (two 32-bit multiplies, a variable shift and an int->float
// conversion per draw, six hashes per sample).  Same structure - a 32-bit LCG per (pixel,
// absolute sample index, seed) whose odd increment selects the stream - but keyed with two
// hashes and read out as the top 23 bits of the state placed in the mantissa of [1, 2).
RTX_DEV uint32_t rng_pixel_key(uint32_t pixel, uint64_t seed) {
    const uint32_t k = rtx_hash32((uint32_t)seed ^ rtx_hash32((uint32_t)(seed >> 32) ^ 0x9e3779b9u));  // wave-uniform
    return rtx_hash32(pixel ^ k);
}
RTX_DEV void rng_init_keyed(RtxRng* r, uint32_t pixel_key, uint32_t sample) {
    r->state = rtx_hash32(sample ^ pixel_key);
    r->inc = (pixel_key << 1) | 1u;
}
RTX_DEV void rng_init(RtxRng* r, uint32_t pixel, uint32_t sample, uint64_t seed) {
    rng_init_keyed(r, rng_pixel_key(pixel, seed), sample);
}
RTX_DEV float rng_nextf(RtxRng* r) {
    const uint32_t s = r->state * 747796405u + r->inc;
    r->state = s;
    return __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, s, 9u)) - 1.0f;  // 0x3f800000 | (s >> 9)
}
#endif

struct HitRec {
    float t;
    V3 p, n;
    uint32_t prim;
    // material of the winning primitive (PrimInfo)
    uint32_t bits;     // kInfo*
    uint32_t texture;
    float param;
    V3 color;
    float u, v;  // texture coordinates; only valid when asked for or when the texture needs them
};
RTX_DEV uint32_t rec_mat_kind(const HitRec& r) { return (r.bits >> kInfoMatShift) & 3u; }

// What the scene holds, for kernels compiled for one scene (rtx_special.cpp defines RTX_SPECIAL_FEATURES):
// a test for something the scene does not have - a textured surface, a fuzzy metal, a sphere among the
// lights - is then decided at compile time and the code behind it goes.  Kernels that serve any scene
// see every feature.
constexpr uint32_t kFeatTextured = 1u, kFeatNeedsUV = 2u, kFeatMetal = 4u, kFeatFuzz = 8u, kFeatDielectric = 16u,
                   kFeatLambertian = 32u, kFeatEmitter = 64u, kFeatLightRect = 128u, kFeatLightSphere = 256u,
                   kFeatLightOther = 512u;
#ifdef RTX_SPECIAL_FEATURES
constexpr uint32_t kSceneFeatures = RTX_SPECIAL_FEATURES;
#else
constexpr uint32_t kSceneFeatures = 0xffffffffu;
#endif
constexpr bool scene_has(uint32_t feature) { return (kSceneFeatures & feature) != 0u; }

// ---------------------------------------------------------------------------
// primitive tests.  Both return the accepted t or leave `t_out` untouched.
// ---------------------------------------------------------------------------
template <int I>
RTX_DEV float comp(V3 a) { return I == 0 ? a.x : (I == 1 ? a.y : a.z); }

// Per-ray constants of one group's frame (fast mode only; empty in deterministic mode).
struct RayAux {
#if !RTX_DET
    V3 inv_d;  // 1 / d
    V3 noi;    // -o / d, so that (k - o) / d = k * inv_d + noi is one fma
#endif
};
RTX_DEV RayAux make_aux(V3 o, V3 d) {
    RayAux a;
#if !RTX_DET
    a.inv_d = v3(frcp(d.x), frcp(d.y), frcp(d.z));
    a.noi = v3(-o.x * a.inv_d.x, -o.y * a.inv_d.y, -o.z * a.inv_d.z);
#endif
    return a;
}

// Rect::hit, Rect.cpp:16-25, for one axis type (XI, YI in-plane, ZI normal).  In
// deterministic mode a NaN t or NaN x / y passes both rejection tests exactly as in the
// reference (the comparisons are written as rejections).
template <int XI, int YI, int ZI>
RTX_DEV bool rect_axis(float x0, float x1, float y0, float y1, float k, V3 o, V3 d, const RayAux& aux,
                       float t0, float t1, float& t_out) {
#if RTX_DET
    float t = (k - comp<ZI>(o)) / comp<ZI>(d);
    if (t < t0 || t > t1) return false;
    float x = comp<XI>(o) + t * comp<XI>(d);
    float y = comp<YI>(o) + t * comp<YI>(d);
    if (x < x0 || x > x1 || y < y0 || y > y1) return false;
#else
    float t = __builtin_fmaf(k, comp<ZI>(aux.inv_d), comp<ZI>(aux.noi));
    float x = __builtin_fmaf(t, comp<XI>(d), comp<XI>(o));
    float y = __builtin_fmaf(t, comp<YI>(d), comp<YI>(o));
    bool ok = (t >= t0) & (t <= t1) & (x >= x0) & (x <= x1) & (y >= y0) & (y <= y1);
    if (!ok) return false;
#endif
    t_out = t;
    return true;
}

// Rect::hit with the axis type taken from the record.  `kind` is wave-uniform on the
// linear-scan path, so the switch is a scalar branch.
RTX_DEV bool rect_test(uint32_t kind, float x0, float x1, float y0, float y1, float k, V3 o, V3 d,
                       const RayAux& aux, float t0, float t1, float& t_out) {
    if (kind == RTX_PRIM_RECT_XY) return rect_axis<0, 1, 2>(x0, x1, y0, y1, k, o, d, aux, t0, t1, t_out);
    if (kind == RTX_PRIM_RECT_XZ) return rect_axis<0, 2, 1>(x0, x1, y0, y1, k, o, d, aux, t0, t1, t_out);
    return rect_axis<1, 2, 0>(x0, x1, y0, y1, k, o, d, aux, t0, t1, t_out);
}

// Sphere::hit, Sphere.cpp:8-33 (roots only; p / n are derived by the caller).
RTX_DEV bool sphere_test(float cx, float cy, float cz, float radius, V3 o, V3 d, float t0, float t1,
                         float& t_out) {
    V3 oc = o - v3(cx, cy, cz);
    float a = dot(d, d);
    float b = 2.0f * dot(oc, d);
    float c = dot(oc, oc) - radius * radius;
#if RTX_DET
    float D = b * b - (4.0f * a) * c;
#else
    float D = rounded_product(b, b) - rounded_product(4.0f * a, c);  // see sphere_discriminant
#endif
#if RTX_DET
    if (D > 0.0f) {
        float root = fsqrt(D);
        float temp = (-b - root) / (2.0f * a);
        if (temp < t1 && temp > t0) { t_out = temp; return true; }
        temp = (-b + root) / (2.0f * a);
        if (temp < t1 && temp > t0) { t_out = temp; return true; }
    }
    return false;
#else
    // branch-free: a negative discriminant makes the roots NaN and every comparison false
    float root = fsqrt(D);
    float inv2a = frcp(2.0f * a);
    float ta = (-b - root) * inv2a;
    float tb = (-b + root) * inv2a;
    bool oka = (D > 0.0f) & (ta < t1) & (ta > t0);
    bool okb = (D > 0.0f) & (tb < t1) & (tb > t0);
    if (!(oka | okb)) return false;
    t_out = oka ? ta : tb;
    return true;
#endif
}

RTX_DEV uint32_t prim_kind(const Prim& P) { return P.kind_flip & 3u; }
RTX_DEV bool prim_flip(const Prim& P) { return (P.kind_flip & 4u) != 0u; }

// One primitive against a ray that is already in the primitive's frame.
RTX_DEV bool prim_test(const Prim& P, V3 o, V3 d, const RayAux& aux, float t0, float t1, float& t) {
    const uint32_t kind = prim_kind(P);
    if (kind == RTX_PRIM_SPHERE) return sphere_test(P.a, P.b, P.c, P.d, o, d, t0, t1, t);
    return rect_test(kind, P.a, P.b, P.c, P.d, P.e, o, d, aux, t0, t1, t);
}

// ---------------------------------------------------------------------------
// transform chains
// ---------------------------------------------------------------------------

// Bring a ray into a group's frame: Translate::hit (Translate.cpp:7) subtracts the offset
// from the origin, Rotate::hit (Rotate.cpp:7-9) rotates origin and direction by the
// conjugate quaternion; outermost wrapper first.  kUniform: the group is the same in every
// lane (scalar loads).
template <bool kUniform>
RTX_DEV void enter_group(const SceneView& sc, uint32_t step_begin, uint32_t step_end, V3& ro, V3& rd) {
    for (uint32_t s = step_begin; s < step_end; ++s) {
        const Step S = kUniform ? load_uniform(sc.steps, s) : sc.steps[s];
        if (S.kind == RTX_STEP_ROTATE) {
            ro = qrotate(-S.x, -S.y, -S.z, S.w, ro);
            rd = qrotate(-S.x, -S.y, -S.z, S.w, rd);
        } else {
            ro = ro - v3(S.x, S.y, S.z);
        }
    }
}

// Texture coordinates of a rect hit (Rect.cpp:27-28) from the in-plane coordinates of the
// hit point in the rect's frame, and of a sphere hit (get_sphere_uv, Util.h:78-83) from the
// unit normal in the sphere's frame, before FlipNormals.
RTX_DEV void rect_uv(uint32_t kind, float x0, float x1, float y0, float y1, V3 ro, V3 rd, float t, float& u,
                     float& v) {
    const int xi = kind == RTX_PRIM_RECT_YZ ? 1 : 0, yi = kind == RTX_PRIM_RECT_XY ? 1 : 2;
    const float ox = xi == 0 ? ro.x : ro.y, dx = xi == 0 ? rd.x : rd.y;
    const float oy = yi == 1 ? ro.y : ro.z, dy = yi == 1 ? rd.y : rd.z;
    float x = ox + t * dx;
    float y = oy + t * dy;
    u = fdiv(x - x0, x1 - x0);
    v = fdiv(y - y0, y1 - y0);
}
RTX_DEV void sphere_uv(V3 n, float& u, float& v) {
    constexpr float kPi = 3.14159265359f;  // PI, Util.h:2
    float phi = rtx_atan2f(n.z, n.x);
    float theta = rtx_asinf_with_root(n.y, __builtin_sqrtf((1.0f - n.y) * (1.0f + n.y)));
    u = 1.0f - fdiv(phi + kPi, 2.0f * kPi);
    v = fdiv(theta + fdiv(kPi, 2.0f), kPi);
}

// Complete the record of the winning primitive (per lane).  In the reference the record is
// filled in the primitive's frame (Rect.cpp:27-32, Sphere.cpp:17-20) and carried out through
// the wrappers innermost first: Rotate::hit (Rotate.cpp:12-13) rotates p and n,
// Translate::hit (Translate.cpp:9) offsets p, FlipNormals::hit (FlipNormals.cpp:8) negates n.
RTX_DEV void finish_hit(const SceneView& sc, uint32_t prim, float t, V3 o, V3 d, bool want_uv, HitRec& rec) {
    const PrimInfo I = sc.info[prim];  // two independent loads, one wait
    const Prim W = sc.prims[prim];
    const uint32_t kind = I.bits & kInfoKindMask;
    const bool chained = I.step_end > I.step_begin;
    const bool uv = want_uv || (I.bits & kInfoNeedsUV);
    V3 ro = o, rd = d;
#if RTX_DET
    if (chained) enter_group<false>(sc, I.step_begin, I.step_end, ro, rd);
    V3 p = ro + rd * t;  // Ray::at, Ray.h:12
    V3 pl = p;
#else
    // rotations and translations keep t, so the world-space point needs no way back; the
    // local point is only needed for a sphere's normal
    V3 p = o + d * t;
    V3 pl = p;
    if (chained && (kind == RTX_PRIM_SPHERE || uv)) {
        enter_group<false>(sc, I.step_begin, I.step_end, ro, rd);
        pl = ro + rd * t;
    }
#endif
    V3 n;
    rec.u = rec.v = 0.0f;
    if (kind == RTX_PRIM_SPHERE) {
#if RTX_DET
        n = vdivs(pl - v3(W.a, W.b, W.c), W.d);  // Sphere.cpp:19
#else
        n = (pl - v3(W.a, W.b, W.c)) * frcp(W.d);
#endif
        if (uv) sphere_uv(n, rec.u, rec.v);  // Sphere.cpp:21
    } else {
        n = v3(kind == RTX_PRIM_RECT_YZ ? 1.0f : 0.0f, kind == RTX_PRIM_RECT_XZ ? 1.0f : 0.0f,
               kind == RTX_PRIM_RECT_XY ? 1.0f : 0.0f);
        if (uv) rect_uv(kind, W.a, W.b, W.c, W.d, ro, rd, t, rec.u, rec.v);
    }
    if (I.bits & kInfoFlip) n = -n;
    if (chained) {
        for (uint32_t s = I.step_end; s > I.step_begin; --s) {
            const Step S = sc.steps[s - 1];
            if (S.kind == RTX_STEP_ROTATE) {
#if RTX_DET
                p = qrotate(S.x, S.y, S.z, S.w, p);
#endif
                n = qrotate(S.x, S.y, S.z, S.w, n);
            } else {
#if RTX_DET
                p = p + v3(S.x, S.y, S.z);
#endif
            }
        }
    }
    rec.t = t;
    rec.p = p;
    rec.n = n;
    rec.prim = prim;
    rec.bits = I.bits;
    rec.texture = I.texture;
    rec.param = I.param;
    rec.color = v3(I.r, I.g, I.b);
}

// ---------------------------------------------------------------------------
// nearest hit: linear scan in list order (ShapeList::hit, ShapeList.cpp:5-17)
// ---------------------------------------------------------------------------
// Every lane of the wave walks the same groups and primitives, so the records are fetched
// with scalar loads.  Only (t, primitive) of the current winner is tracked during the scan;
// the record is completed once, after the scan, for the final winner.
#if RTX_DET
// Deterministic mode: list order, one generic loop (the order decides ties and the fate of
// NaN candidates, both of which must come out as in the reference).  The next record is
// requested before the current one is tested.
RTX_DEV bool query_linear(const SceneView& sc, V3 o, V3 d, float t0, float t1, float& t_hit, uint32_t& prim_hit) {
    float closest = t1;
    uint32_t best = kNone;
    for (uint32_t g = 0; g < sc.n_groups; ++g) {
        const Group G = load_uniform(sc.groups, g);
        V3 ro = o, rd = d;
        enter_group<true>(sc, G.step_begin, G.step_end, ro, rd);
        const RayAux aux = make_aux(ro, rd);
        Prim cur = load_uniform(sc.prims, G.prim_begin);
        for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) {
            const Prim nxt = load_uniform(sc.prims, p + 1u);  // the array ends with a pad record
            float t;
            if (prim_test(cur, ro, rd, aux, t0, closest, t)) {
                closest = t;
                best = p;
            }
            cur = nxt;
        }
    }
    t_hit = closest;
    prim_hit = best;
    return best != kNone;
}
RTX_DEV uint32_t original_prim(const SceneView&, uint32_t prim, bool) { return prim; }
#else
// Fast mode: the bucketed table (ScanGroup / ScanRec): per group one composed affine map and
// four homogeneous loops (XY, XZ, YZ rects, spheres) whose bodies are branch-free.
// Complete the record of the winner of a fast-mode query (per lane): one 64-byte record in
// world space, no transform.  `slot` is a position of scan_info / scan_orig.
template <bool kBvh>
RTX_DEV void finish_scan_hit(const SceneView& sc, uint32_t slot, float t, V3 o, V3 d, bool want_uv, HitRec& rec) {
    const ScanHit I = at32(sc.scan_info, slot);
    const V3 p = o + d * t;
    const V3 pc = p - v3(I.c);
    rec.t = t;
    rec.p = p;
    rec.n = v3(__builtin_fmaf(pc.x, I.inv_r, I.n0[0]), __builtin_fmaf(pc.y, I.inv_r, I.n0[1]),
               __builtin_fmaf(pc.z, I.inv_r, I.n0[2]));
    rec.prim = slot;
    rec.bits = I.bits;
    rec.texture = I.texture;
    rec.param = I.param;
    rec.color = v3(I.r, I.g, I.b);
    rec.u = rec.v = 0.0f;
    if (want_uv || (scene_has(kFeatNeedsUV) && (I.bits & kInfoNeedsUV))) {  // texture coordinates live in the primitive's frame
        const Prim W = sc.prims[sc.scan_orig[slot]];
        const uint32_t kind = I.bits & kInfoKindMask;
        V3 pl = p;
        if (I.chained) {
            if (kBvh) {  // no composed map per group there: walk the step chain
                V3 unused = d;
                const Group G = sc.groups[I.group];
                enter_group<false>(sc, G.step_begin, G.step_end, pl, unused);
            } else {
                const float* m = sc.scan_groups[I.group].m;
                pl = v3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[9], m[3] * p.x + m[4] * p.y + m[5] * p.z + m[10],
                        m[6] * p.x + m[7] * p.y + m[8] * p.z + m[11]);
            }
        }
        if (kind == RTX_PRIM_SPHERE) {
            V3 nl = (pl - v3(W.a, W.b, W.c)) * frcp(W.d);  // before FlipNormals (Sphere.cpp:19-21)
            sphere_uv(nl, rec.u, rec.v);
        } else {
            rect_uv(kind, W.a, W.b, W.c, W.d, pl, v3(0.0f, 0.0f, 0.0f), 0.0f, rec.u, rec.v);
        }
    }
}

// One rect of the bucketed table against every lane's ray.  The six interval tests are
// v_cmpx instructions: each one clears the lanes that fail from EXEC, so no mask has to be
// combined on the scalar unit and the two updates are plain moves under the surviving lanes.
template <int XI, int YI, int ZI>
RTX_DEV void scan_rect(const ScanRec& R, uint32_t pos, V3 ro, V3 rd, const RayAux& aux, float t0, float& closest,
                       uint32_t& best) {
    float t, x, y;
    unsigned long long saved;
    // not volatile: the statement touches no memory, and a volatile one would stop the
    // compiler from proving the record loads around it uniform and read-only (scalar loads)
    asm("v_fma_f32 %[t], %[k], %[inv], %[noi]\n\t"
        "v_fma_f32 %[x], %[t], %[dx], %[ox]\n\t"
        "v_fma_f32 %[y], %[t], %[dy], %[oy]\n\t"
        "s_mov_b64 %[saved], exec\n\t"
        "v_cmpx_le_f32 vcc, %[t0], %[t]\n\t"
        "v_cmpx_ge_f32 vcc, %[closest], %[t]\n\t"
        "v_cmpx_le_f32 vcc, %[x0], %[x]\n\t"
        "v_cmpx_ge_f32 vcc, %[x1], %[x]\n\t"
        "v_cmpx_le_f32 vcc, %[y0], %[y]\n\t"
        "v_cmpx_ge_f32 vcc, %[y1], %[y]\n\t"
        "v_mov_b32 %[closest], %[t]\n\t"
        "v_mov_b32 %[best], %[pos]\n\t"
        "s_mov_b64 exec, %[saved]"
        : [t] "=&v"(t), [x] "=&v"(x), [y] "=&v"(y), [saved] "=&s"(saved), [closest] "+v"(closest), [best] "+v"(best)
        : [k] "s"(R.e), [inv] "v"(comp<ZI>(aux.inv_d)), [noi] "v"(comp<ZI>(aux.noi)), [dx] "v"(comp<XI>(rd)),
          [ox] "v"(comp<XI>(ro)), [dy] "v"(comp<YI>(rd)), [oy] "v"(comp<YI>(ro)), [t0] "s"(t0), [x0] "s"(R.a),
          [x1] "s"(R.b), [y0] "s"(R.c), [y1] "s"(R.d), [pos] "s"(pos)
        : "vcc");
}

template <int XI, int YI, int ZI>
RTX_DEV void scan_rects(const SceneView& sc, uint32_t& pos, uint32_t count, V3 ro, V3 rd, const RayAux& aux,
                        float t0, float& closest, uint32_t& best) {
    for (uint32_t k = 0; k < count; ++k, ++pos) {
        const ScanRec R = load_uniform(sc.scan_recs, pos);
        scan_rect<XI, YI, ZI>(R, pos, ro, rd, aux, t0, closest, best);
    }
}

// Complete boxes (six rects that are the faces of one axis-aligned box, Box.cpp:11-29) by a
// slab test: the ray meets the planes of axis a at tl_a, th_a; it is inside all three slabs
// for t in [max near_a, min far_a].  Of a closed convex box it can only hit the face it
// enters through or, when that lies behind t0 (origin inside or on the box), the face it
// leaves through; each rect's own t is the same fma as in scan_rect, its bounds tests are
// implied by the other two slabs.  The face is recovered from which slab produced t.
RTX_DEV void scan_boxes(const SceneView& sc, uint32_t& pos, uint32_t count, V3 ro, const RayAux& aux, float t0,
                        float& closest, uint32_t& best) {
    // a direction component of exactly zero makes inv_d infinite and k * inv_d + noi NaN, which
    // min / max would silently drop: keep the reciprocal finite, then a parallel ray is inside
    // the slab for all t or for none
    constexpr float kBig = 1.0e30f;
    const V3 inv = v3(__builtin_amdgcn_fmed3f(aux.inv_d.x, -kBig, kBig), __builtin_amdgcn_fmed3f(aux.inv_d.y, -kBig, kBig),
                      __builtin_amdgcn_fmed3f(aux.inv_d.z, -kBig, kBig));
    const V3 noi = v3(-ro.x * inv.x, -ro.y * inv.y, -ro.z * inv.z);
    for (uint32_t k = 0; k < count; ++k, pos += kBoxSlots) {
        const ScanRec R = load_uniform(sc.scan_recs, pos);
        const float lx = __builtin_fmaf(R.a, inv.x, noi.x), hx = __builtin_fmaf(R.d, inv.x, noi.x);
        const float ly = __builtin_fmaf(R.b, inv.y, noi.y), hy = __builtin_fmaf(R.e, inv.y, noi.y);
        const float lz = __builtin_fmaf(R.c, inv.z, noi.z), hz = __builtin_fmaf(R.f, inv.z, noi.z);
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(lx, hx), __builtin_fminf(ly, hy)), __builtin_fminf(lz, hz));
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(lx, hx), __builtin_fmaxf(ly, hy)), __builtin_fmaxf(lz, hz));
        const float t = tn >= t0 ? tn : tf;
        const bool ok = (tn <= tf) & (t >= t0) & (t <= closest);
        // the face is the plane whose distance t is (slot order +z -z +y -y +x -x)
        // (one select per plane: as a chain of nested conditionals it became four nested branches)
        uint32_t face = 0u;
        face = t == lz ? 1u : face;
        face = t == hy ? 2u : face;
        face = t == ly ? 3u : face;
        face = t == hx ? 4u : face;
        face = t == lx ? 5u : face;
        closest = ok ? t : closest;
        best = ok ? pos + face : best;
    }
}

// The scan of ONE scene as straight-line code with its records as literal operands.  rtx_upload_scene
// generates such a scan for the scene at hand (rtx_special.cpp: one RTX_SPECIAL_GROUP_* / RTX_SPECIAL_RECT /
// RTX_SPECIAL_SPHERE / RTX_SPECIAL_BOX line per group and record, in the table's order, as the macro
// RTX_SPECIAL_SCAN) and compiles this file around it at run time.  Same tests, same arithmetic, same
// winners as the table-driven scan below; what goes is what the table costs: a vector instruction with a
// scalar-register operand issues at half the rate of one with a literal, and every record costs a scalar
// load, a wait and the loop's bookkeeping (Cornell box: 104.7 -> 114.4 Grays/s).  Arguments are the bits
// of the fp32 values.
#define RTX_SPECIAL_F(bits) __builtin_bit_cast(float, (uint32_t)(bits))
// lo <= v <= hi as two v_cmpx, and for lo == +0 and hi > 0 as one: the bits of a float in [0, hi] are, read
// as an unsigned integer, at most those of hi, and those of a negative number or a NaN are above them
// (only -0.0 ends up on the other side of the test)
#define RTX_SPECIAL_RANGE(LO, HI, V) "v_cmpx_le_f32 vcc, %[" LO "], %[" V "]\n\tv_cmpx_ge_f32 vcc, %[" HI "], %[" V "]\n\t"
#define RTX_SPECIAL_RANGE0(LO, HI, V) "v_cmpx_ge_u32 vcc, %[" HI "], %[" V "]\n\t"
#define RTX_SPECIAL_RECT(XI, YI, ZI, XR, YR, X0, X1, Y0, Y1, K, POS)                                                \
    do {                                                                                                             \
        float t_, x_, y_;                                                                                            \
        unsigned long long saved_;                                                                                   \
        asm("v_fmamk_f32 %[t], %[inv], %[k], %[noi]\n\t"                                                             \
            "v_fma_f32 %[x], %[t], %[dx], %[ox]\n\t"                                                                 \
            "v_fma_f32 %[y], %[t], %[dy], %[oy]\n\t"                                                                 \
            "s_mov_b64 %[saved], exec\n\t"                                                                           \
            "v_cmpx_le_f32 vcc, 0x3a83126f, %[t]\n\t" /* kTMin */                                                    \
            "v_cmpx_ge_f32 vcc, %[closest], %[t]\n\t"                                                                \
            XR("x0", "x1", "x") YR("y0", "y1", "y")                                                                  \
            "v_mov_b32 %[closest], %[t]\n\t"                                                                         \
            "v_mov_b32 %[best], %[pos]\n\t"                                                                          \
            "s_mov_b64 exec, %[saved]"                                                                               \
            : [t] "=&v"(t_), [x] "=&v"(x_), [y] "=&v"(y_), [saved] "=&s"(saved_), [closest] "+v"(closest_), [best] "+v"(best_) \
            : [k] "i"(K), [inv] "v"(comp<ZI>(aux_.inv_d)), [noi] "v"(comp<ZI>(aux_.noi)), [dx] "v"(comp<XI>(rd_)),       \
              [ox] "v"(comp<XI>(ro_)), [dy] "v"(comp<YI>(rd_)), [oy] "v"(comp<YI>(ro_)), [x0] "i"(X0),                    \
              [x1] "i"(X1), [y0] "i"(Y0), [y1] "i"(Y1), [pos] "i"(POS)                                                \
            : "vcc");                                                                                                \
    } while (0)
// (a group's frame: o_, d_ are the ray in world space; the maps are those of ScanGroup)
#define RTX_SPECIAL_GROUP_IDENTITY() const V3 ro_ = o_, rd_ = d_; const RayAux aux_ = make_aux(ro_, rd_)
#define RTX_SPECIAL_GROUP_ROT_Y(M0, M2, M6, M8, M9, M10, M11)                                                        \
    const V3 ro_ = v3(RTX_SPECIAL_F(M0) * o_.x + RTX_SPECIAL_F(M2) * o_.z + RTX_SPECIAL_F(M9), o_.y + RTX_SPECIAL_F(M10),  \
                      RTX_SPECIAL_F(M6) * o_.x + RTX_SPECIAL_F(M8) * o_.z + RTX_SPECIAL_F(M11)),                       \
             rd_ = v3(RTX_SPECIAL_F(M0) * d_.x + RTX_SPECIAL_F(M2) * d_.z, d_.y, RTX_SPECIAL_F(M6) * d_.x + RTX_SPECIAL_F(M8) * d_.z); \
    const RayAux aux_ = make_aux(ro_, rd_)
#define RTX_SPECIAL_GROUP_AFFINE(M0, M1, M2, M3, M4, M5, M6, M7, M8, M9, M10, M11)                                    \
    const V3 ro_ = v3(RTX_SPECIAL_F(M0) * o_.x + RTX_SPECIAL_F(M1) * o_.y + RTX_SPECIAL_F(M2) * o_.z + RTX_SPECIAL_F(M9),  \
                      RTX_SPECIAL_F(M3) * o_.x + RTX_SPECIAL_F(M4) * o_.y + RTX_SPECIAL_F(M5) * o_.z + RTX_SPECIAL_F(M10), \
                      RTX_SPECIAL_F(M6) * o_.x + RTX_SPECIAL_F(M7) * o_.y + RTX_SPECIAL_F(M8) * o_.z + RTX_SPECIAL_F(M11)), \
             rd_ = v3(RTX_SPECIAL_F(M0) * d_.x + RTX_SPECIAL_F(M1) * d_.y + RTX_SPECIAL_F(M2) * d_.z,                    \
                      RTX_SPECIAL_F(M3) * d_.x + RTX_SPECIAL_F(M4) * d_.y + RTX_SPECIAL_F(M5) * d_.z,                    \
                      RTX_SPECIAL_F(M6) * d_.x + RTX_SPECIAL_F(M7) * d_.y + RTX_SPECIAL_F(M8) * d_.z);                   \
    const RayAux aux_ = make_aux(ro_, rd_)
#define RTX_SPECIAL_SPHERES() const float a_ = dot(rd_, rd_), inv_a_ = frcp(a_)
#define RTX_SPECIAL_SPHERE(CX, CY, CZ, RR, POS)                                                                      \
    do { /* as the sphere loop of query_linear */                                                                    \
        const V3 oc = ro_ - v3(RTX_SPECIAL_F(CX), RTX_SPECIAL_F(CY), RTX_SPECIAL_F(CZ));                               \
        const float h = dot(oc, rd_);                                                                                \
        const float c = dot(oc, oc) - RTX_SPECIAL_F(RR);                                                             \
        const float D = sphere_discriminant(h, a_, c);                                                               \
        const float root = fsqrt(D);                                                                                 \
        const float ta = (-h - root) * inv_a_;                                                                       \
        const float tb = (-h + root) * inv_a_;                                                                       \
        const float t = ta > t0_ ? ta : tb;                                                                          \
        const bool ok = (tb > t0_) & (t < closest_);                                                                 \
        closest_ = ok ? t : closest_;                                                                                \
        best_ = ok ? (uint32_t)(POS) : best_;                                                                        \
    } while (0)
#define RTX_SPECIAL_BOXES()                                                                                          \
    const V3 binv_ = v3(__builtin_amdgcn_fmed3f(aux_.inv_d.x, -1.0e30f, 1.0e30f), __builtin_amdgcn_fmed3f(aux_.inv_d.y, -1.0e30f, 1.0e30f), \
                        __builtin_amdgcn_fmed3f(aux_.inv_d.z, -1.0e30f, 1.0e30f));                                   \
    const V3 bnoi_ = v3(-ro_.x * binv_.x, -ro_.y * binv_.y, -ro_.z * binv_.z)
// (a plane at coordinate +0: 0 * binv_ is 0 because binv_ is finite, the distance is bnoi_ itself)
#define RTX_SPECIAL_PLANE(K, INV, NOI) ((uint32_t)(K) == 0u ? (NOI) : __builtin_fmaf(RTX_SPECIAL_F(K), INV, NOI))
#define RTX_SPECIAL_BOX(LX, LY, LZ, HX, HY, HZ, POS)                                                                 \
    do { /* as scan_boxes */                                                                                         \
        const float lx = RTX_SPECIAL_PLANE(LX, binv_.x, bnoi_.x), hx = RTX_SPECIAL_PLANE(HX, binv_.x, bnoi_.x);        \
        const float ly = RTX_SPECIAL_PLANE(LY, binv_.y, bnoi_.y), hy = RTX_SPECIAL_PLANE(HY, binv_.y, bnoi_.y);        \
        const float lz = RTX_SPECIAL_PLANE(LZ, binv_.z, bnoi_.z), hz = RTX_SPECIAL_PLANE(HZ, binv_.z, bnoi_.z);        \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(lx, hx), __builtin_fminf(ly, hy)), __builtin_fminf(lz, hz)); \
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(lx, hx), __builtin_fmaxf(ly, hy)), __builtin_fmaxf(lz, hz)); \
        const float t = tn >= t0_ ? tn : tf;                                                                         \
        const bool ok = (tn <= tf) & (t >= t0_) & (t <= closest_);                                                   \
        uint32_t face = 0u;                                                                                          \
        face = t == lz ? 1u : face;                                                                                  \
        face = t == hy ? 2u : face;                                                                                  \
        face = t == ly ? 3u : face;                                                                                  \
        face = t == hx ? 4u : face;                                                                                  \
        face = t == lx ? 5u : face;                                                                                  \
        closest_ = ok ? t : closest_;                                                                                \
        best_ = ok ? (uint32_t)(POS) + face : best_;                                                                 \
    } while (0)

RTX_DEV bool query_linear(const SceneView& sc, V3 o, V3 d, float t0, float t1, float& t_hit, uint32_t& prim_hit) {
    float closest = t1;
    uint32_t best = kNone;
#ifdef RTX_SPECIAL_SCAN
    static_assert(__builtin_bit_cast(uint32_t, kTMin) == 0x3a83126fu, "the literal in RTX_SPECIAL_RECT");
    if (t0 == kTMin) {  // (what the trace kernels ask; the comparison folds away once the call is inlined)
        const V3 o_ = o, d_ = d;
        constexpr float t0_ = kTMin;  // (a constant for the compiler, not the register that compared equal to it)
        float closest_ = closest;
        uint32_t best_ = best;
        RTX_SPECIAL_SCAN
        t_hit = closest_;
        prim_hit = best_;
        return best_ != kNone;
    }
#endif
    uint32_t g = 0u;
    do {  // (the table of an uploaded scene has at least one group: no test before the first)
        const ScanGroup G = load_uniform(sc.scan_groups, g);
        V3 ro = o, rd = d;
        if (G.rot_y) {  // the products with the exact zeros and the one of the map left out
            ro = v3(G.m[0] * o.x + G.m[2] * o.z + G.m[9], o.y + G.m[10], G.m[6] * o.x + G.m[8] * o.z + G.m[11]);
            rd = v3(G.m[0] * d.x + G.m[2] * d.z, d.y, G.m[6] * d.x + G.m[8] * d.z);
        } else if (!G.identity) {
            ro = v3(G.m[0] * o.x + G.m[1] * o.y + G.m[2] * o.z + G.m[9],
                    G.m[3] * o.x + G.m[4] * o.y + G.m[5] * o.z + G.m[10],
                    G.m[6] * o.x + G.m[7] * o.y + G.m[8] * o.z + G.m[11]);
            rd = v3(G.m[0] * d.x + G.m[1] * d.y + G.m[2] * d.z, G.m[3] * d.x + G.m[4] * d.y + G.m[5] * d.z,
                    G.m[6] * d.x + G.m[7] * d.y + G.m[8] * d.z);
        }
        const RayAux aux = make_aux(ro, rd);
        uint32_t pos = G.first;
        scan_rects<0, 1, 2>(sc, pos, G.n_xy, ro, rd, aux, t0, closest, best);
        scan_rects<0, 2, 1>(sc, pos, G.n_xz, ro, rd, aux, t0, closest, best);
        scan_rects<1, 2, 0>(sc, pos, G.n_yz, ro, rd, aux, t0, closest, best);
        if (G.n_sph) {
            // roots of a t^2 + 2 h t + c = 0 with h = oc.d (Sphere.cpp:9-15 with b = 2h)
            const float a = dot(rd, rd);
            const float inv_a = frcp(a);
            for (uint32_t k = 0; k < G.n_sph; ++k, ++pos) {
                const ScanRec R = load_uniform(sc.scan_recs, pos);
                V3 oc = ro - v3(R.a, R.b, R.c);
                float h = dot(oc, rd);
                float c = dot(oc, oc) - R.e;
                float D = sphere_discriminant(h, a, c);
                float root = fsqrt(D);  // NaN for a negative discriminant: every test below fails
                float ta = (-h - root) * inv_a;
                float tb = (-h + root) * inv_a;
                // ta <= tb: the near root if it lies beyond t0, else the far one; a near root
                // at or beyond `closest` rules the far one out too (Sphere.cpp:14-31)
                // (a negative discriminant makes both roots NaN and every comparison false; with
                // ta <= tb, "the chosen root lies beyond t0" is "tb lies beyond t0")
                float t = ta > t0 ? ta : tb;
                bool ok = (tb > t0) & (t < closest);
                closest = ok ? t : closest;
                best = ok ? pos : best;
            }
        }
        if (G.n_box) scan_boxes(sc, pos, G.n_box, ro, aux, t0, closest, best);
    } while (++g < sc.n_scan_groups);
    t_hit = closest;
    prim_hit = best;
    return best != kNone;
}
// The fast linear scan reports positions of the bucketed table.
RTX_DEV uint32_t original_prim(const SceneView& sc, uint32_t prim, bool) { return sc.scan_orig[prim]; }
#endif

// ---------------------------------------------------------------------------
// nearest hit: BVH over groups, per-lane stack
// ---------------------------------------------------------------------------
//
// The linear scan accepts a Rect at t == closest but a Sphere only at
// t < closest, so among primitives that tie for the minimum t the reference
// returns the LAST Rect in list order if any Rect ties, else the FIRST Sphere.
// Traversal order differs from list order, so ties are resolved explicitly by
// that rule; everything else (each primitive's own candidate t) does not depend
// on the order of the scan.
#ifndef RTX_TRACE_WAVES_BVH
#define RTX_TRACE_WAVES_BVH 4
#endif
constexpr int kBvhStack = kBvhStackEntries;  // one deferred child per level

RTX_DEV bool tie_wins(uint32_t cand, bool cand_is_rect, uint32_t cur, bool cur_is_rect) {
    if (cand_is_rect && cur_is_rect) return cand > cur;
    if (cand_is_rect != cur_is_rect) return cand_is_rect;
    return cand < cur;
}

// The nodes' boxes are 16-bit coordinates q on the grid origin + q * scale (rounded outwards
// by the builder, which also pads the boxes by far more than the rounding of these products).
// Per ray: t(q) = (origin + q * scale - o) * inv = q * A + B with A = scale * inv and
// B = origin * inv + noi, noi = -o * inv, so a plane distance is one conversion and one fma.
// inv must be finite (the caller clamps it): a ray (all but) parallel to a slab is then inside
// it for every t or for none.
struct BvhRay {
    V3 A, B;
};
RTX_DEV BvhRay bvh_ray(const SceneView& sc, V3 inv, V3 noi) {
    BvhRay r;
    r.A = v3(sc.bvh_scale[0] * inv.x, sc.bvh_scale[1] * inv.y, sc.bvh_scale[2] * inv.z);
    r.B = v3(__builtin_fmaf(sc.bvh_origin[0], inv.x, noi.x), __builtin_fmaf(sc.bvh_origin[1], inv.y, noi.y),
             __builtin_fmaf(sc.bvh_origin[2], inv.z, noi.z));
    return r;
}
// One child's box: w0 = lo.x | hi.x << 16, w1 = lo.y | hi.y << 16, w2 = lo.z | hi.z << 16.
// Conservative; returns the entry distance.
RTX_DEV bool slab(uint32_t w0, uint32_t w1, uint32_t w2, const BvhRay& r, float t0, float t1, float& t_near) {
    const float lx = (float)(w0 & 0xffffu), ly = (float)(w1 & 0xffffu), lz = (float)(w2 & 0xffffu);
    const float hx = (float)(w0 >> 16), hy = (float)(w1 >> 16), hz = (float)(w2 >> 16);
    float tx0 = __builtin_fmaf(lx, r.A.x, r.B.x), tx1 = __builtin_fmaf(hx, r.A.x, r.B.x);
    float ty0 = __builtin_fmaf(ly, r.A.y, r.B.y), ty1 = __builtin_fmaf(hy, r.A.y, r.B.y);
    float tz0 = __builtin_fmaf(lz, r.A.z, r.B.z), tz1 = __builtin_fmaf(hz, r.A.z, r.B.z);
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), t0));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), t1));
    t_near = tn;
    return tn <= tf;
}
#if !RTX_DET
// The same test when the ray knows which plane of each axis it meets first.  The five low mantissa
// bits of B hold 16 where the ray runs towards -axis and 0 where it runs towards +axis
// (bvh_ray_signed; 31 ulp of B move a plane by a few millionths of the distance between the ray's
// origin and the grid's, a hundredth of the padding of the boxes): a rotate of the word by that
// amount - v_alignbit_b32 reads the amount from just those bits - leaves near | far << 16, and the
// three min / max pairs of the plain test are gone.
RTX_DEV BvhRay bvh_ray_signed(const SceneView& sc, V3 inv, V3 noi) {
    BvhRay r = bvh_ray(sc, inv, noi);
    r.B.x = __uint_as_float((__float_as_uint(r.B.x) & ~31u) | ((__float_as_uint(r.A.x) >> 27) & 16u));
    r.B.y = __uint_as_float((__float_as_uint(r.B.y) & ~31u) | ((__float_as_uint(r.A.y) >> 27) & 16u));
    r.B.z = __uint_as_float((__float_as_uint(r.B.z) & ~31u) | ((__float_as_uint(r.A.z) >> 27) & 16u));
    return r;
}
RTX_DEV bool slab_signed(uint32_t w0, uint32_t w1, uint32_t w2, const BvhRay& r, float t0, float t1, float& t_near) {
    const uint32_t x = __builtin_amdgcn_alignbit(w0, w0, __float_as_uint(r.B.x));
    const uint32_t y = __builtin_amdgcn_alignbit(w1, w1, __float_as_uint(r.B.y));
    const uint32_t z = __builtin_amdgcn_alignbit(w2, w2, __float_as_uint(r.B.z));
    const float nx = __builtin_fmaf((float)(x & 0xffffu), r.A.x, r.B.x), fx = __builtin_fmaf((float)(x >> 16), r.A.x, r.B.x);
    const float ny = __builtin_fmaf((float)(y & 0xffffu), r.A.y, r.B.y), fy = __builtin_fmaf((float)(y >> 16), r.A.y, r.B.y);
    const float nz = __builtin_fmaf((float)(z & 0xffffu), r.A.z, r.B.z), fz = __builtin_fmaf((float)(z >> 16), r.A.z, r.B.z);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx, ny), nz), t0);
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx, fy), fz), t1);
    t_near = tn;
    return tn <= tf;
}
#endif
// A node as its eight 32-bit words (two 16-byte loads).
struct NodeWords {
    uint4 a, b;  // a.xyz: child 0's box, a.w | b.xy: child 1's box, b.z / b.w: the children
};
RTX_DEV NodeWords load_node(const SceneView& sc, uint32_t ref) {
    const uint4* p = reinterpret_cast<const uint4*>(&at32(sc.bvh, ref));
    NodeWords n;
    n.a = p[0];
    n.b = p[1];
    return n;
}

// Next deferred subtree of this lane, or 0xffffffff when its stack is empty.  An entry that
// did not fit (the builder bounds the depth, so this does not happen) reads as an empty leaf.
template <int kCols>
RTX_DEV uint32_t pop_ref(uint32_t (*stack)[kCols], uint32_t& sp, uint32_t tid) {
    if (sp == 0u) return 0xffffffffu;
    --sp;
    return sp < (uint32_t)kBvhStack ? stack[sp][tid] : kBvhLeaf;
}


#if !RTX_DET
// One fast-mode leaf record against this lane's ray (inv / noi: the capped reciprocal
// direction and -o * inv of the traversal).  Ties between equal distances are not ordered:
// fast mode only has to be statistically equal.
// A lane's leaf record as two 16-byte loads issued together: read field by field, the type
// comes first and the fields of that type a memory round trip later.
RTX_DEV BvhFastItem load_fast_item(const BvhFastItem* base, uint32_t i) {
    static_assert(sizeof(BvhFastItem) == 32, "two uint4");
    const uint4* q = reinterpret_cast<const uint4*>(&at32(base, i));
    uint4 a = q[0], b = q[1];
    asm("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
    BvhFastItem I;
    I.f[0] = __uint_as_float(a.x); I.f[1] = __uint_as_float(a.y); I.f[2] = __uint_as_float(a.z);
    I.f[3] = __uint_as_float(a.w); I.f[4] = __uint_as_float(b.x);
    I.slot = b.y; I.type_count = b.z; I.map = b.w;
    return I;
}

RTX_DEV void fast_leaf_item(const SceneView& sc, const BvhFastItem& I, V3 o, V3 d, V3 inv, V3 noi, float t0,
                            float& closest, uint32_t& best) {
    const uint32_t type = I.type_count & 0xffu;
    if (type == kBvhItemSphere) {
        // roots of a t^2 + 2 h t + c = 0 with h = oc.d (Sphere.cpp:9-15 with b = 2h)
        const float a = dot(d, d), inv_a = frcp(a);
        const V3 oc = o - v3(I.f[0], I.f[1], I.f[2]);
        const float h = dot(oc, d), c = dot(oc, oc) - I.f[3];
        const float D = sphere_discriminant(h, a, c);
        const float root = fsqrt(D);
        const float ta = (-h - root) * inv_a, tb = (-h + root) * inv_a;
        const float t = ta > t0 ? ta : tb;
        const bool ok = (D > 0.0f) & (t > t0) & (t < closest);
        closest = ok ? t : closest;
        best = ok ? I.slot : best;
    } else if (type <= kBvhItemRectYZ) {
        const bool xy = type == kBvhItemRectXY, yz = type == kBvhItemRectYZ;
        // plane coordinate zi, in-plane xi / yi: XY (0,1,2), XZ (0,2,1), YZ (1,2,0)
        const float iz = xy ? inv.z : (yz ? inv.x : inv.y), nz = xy ? noi.z : (yz ? noi.x : noi.y);
        const float ox = yz ? o.y : o.x, dx = yz ? d.y : d.x;
        const float oy = xy ? o.y : o.z, dy = xy ? d.y : d.z;
        const float t = __builtin_fmaf(I.f[4], iz, nz);
        const float x = __builtin_fmaf(t, dx, ox), y = __builtin_fmaf(t, dy, oy);
        const bool ok = (t >= t0) & (t <= closest) & (x >= I.f[0]) & (x <= I.f[1]) & (y >= I.f[2]) & (y <= I.f[3]);
        closest = ok ? t : closest;
        best = ok ? I.slot : best;
    } else if (type == kBvhItemBox) {
        // the box is [0,1]^3 in the record's frame: plane distances are noi and inv + noi
        const BvhMap M = at32(sc.bvh_maps, I.map);
        const float* m = M.m;
        const V3 ro = v3(m[0] * o.x + m[1] * o.y + m[2] * o.z + m[9], m[3] * o.x + m[4] * o.y + m[5] * o.z + m[10],
                         m[6] * o.x + m[7] * o.y + m[8] * o.z + m[11]);
        const V3 rd = v3(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[3] * d.x + m[4] * d.y + m[5] * d.z,
                         m[6] * d.x + m[7] * d.y + m[8] * d.z);
        constexpr float kBig = 1.0e18f;
        const V3 li = v3(__builtin_amdgcn_fmed3f(frcp(rd.x), -kBig, kBig), __builtin_amdgcn_fmed3f(frcp(rd.y), -kBig, kBig),
                         __builtin_amdgcn_fmed3f(frcp(rd.z), -kBig, kBig));
        const float lx = -ro.x * li.x, ly = -ro.y * li.y, lz = -ro.z * li.z;
        const float hx = lx + li.x, hy = ly + li.y, hz = lz + li.z;
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(lx, hx), __builtin_fminf(ly, hy)), __builtin_fminf(lz, hz));
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(lx, hx), __builtin_fmaxf(ly, hy)), __builtin_fmaxf(lz, hz));
        const float t = tn >= t0 ? tn : tf;
        const bool ok = (tn <= tf) & (t >= t0) & (t <= closest);
        // the face is the plane whose distance t is (slot order +z -z +y -y +x -x)
        const uint32_t face = t == lx ? 5u : (t == hx ? 4u : (t == ly ? 3u : (t == hy ? 2u : (t == lz ? 1u : 0u))));
        closest = ok ? t : closest;
        best = ok ? I.slot + face : best;
    } else {  // any other transformed group: its primitives one by one in the group's frame
        const BvhMap M = at32(sc.bvh_maps, I.map);
        const float* m = M.m;
        const uint32_t first = __float_as_uint(I.f[0]), count = I.type_count >> 8;
        const V3 ro = v3(m[0] * o.x + m[1] * o.y + m[2] * o.z + m[9], m[3] * o.x + m[4] * o.y + m[5] * o.z + m[10],
                         m[6] * o.x + m[7] * o.y + m[8] * o.z + m[11]);
        const V3 rd = v3(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[3] * d.x + m[4] * d.y + m[5] * d.z,
                         m[6] * d.x + m[7] * d.y + m[8] * d.z);
        const RayAux aux = make_aux(ro, rd);
        for (uint32_t k = 0; k < count; ++k) {
            const Prim P = at32(sc.prims, first + k);
            float t;
            if (prim_test(P, ro, rd, aux, t0, closest, t)) {
                closest = t;
                best = I.slot + k;
            }
        }
    }
}
#endif

// Traversal: each lane descends until it holds a leaf (nearer child first, the other one
// deferred on its stack), then the lanes of the wave test their leaves together.  The stack
// lives in LDS, one column per thread (entry e of thread t at [e][t]: conflict-free).
RTX_DEV bool query_bvh(const SceneView& sc, V3 o, V3 d, float t0, float t1, float& t_hit, uint32_t& prim_hit) {
    __shared__ uint32_t s_stack[kBvhStack][256];  // hit_kernel runs 256 threads; trace blocks use a prefix
    const uint32_t tid = threadIdx.x;
    uint32_t sp = 0;
    float closest = t1;
    uint32_t best = kNone;
    bool best_is_rect = false;
    // |1/d| capped: a ray (all but) parallel to a slab is then inside it for every t or for none
    constexpr float kBig = 1.0e18f;
    const V3 inv = v3(__builtin_amdgcn_fmed3f(frcp(d.x), -kBig, kBig), __builtin_amdgcn_fmed3f(frcp(d.y), -kBig, kBig),
                      __builtin_amdgcn_fmed3f(frcp(d.z), -kBig, kBig));
    const V3 noi = v3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    constexpr uint32_t kDone = 0xffffffffu;
    const BvhRay br = bvh_ray(sc, inv, noi);
    uint32_t ref = 0;  // root node
    // The items outside the tree first (every ray tests them, the loop is wave-uniform), then the
    // tree; `big` counts down to the switch.
    uint32_t big = sc.n_bvh_big;
    for (;;) {
        uint32_t first, count;
        if (big != 0u) {
            --big;
            first = sc.bvh_big_first + big;
            count = 1u;
        } else {
            // descend to a leaf
            while (ref != kDone && !(ref & kBvhLeaf)) {
                const NodeWords N = load_node(sc, ref);
                float n0, n1;
                const bool h0 = slab(N.a.x, N.a.y, N.a.z, br, t0, closest, n0);
                const bool h1 = slab(N.a.w, N.b.x, N.b.y, br, t0, closest, n1);
                if (h0 && h1) {
                    const bool first0 = n0 <= n1;
                    if (sp < (uint32_t)kBvhStack) s_stack[sp][tid] = first0 ? N.b.w : N.b.z;
                    ++sp;
                    ref = first0 ? N.b.z : N.b.w;
                } else if (h0 || h1) {
                    ref = h0 ? N.b.z : N.b.w;
                } else {
                    ref = pop_ref(s_stack, sp, tid);
                }
            }
            if (ref == kDone) break;
            // leaf: items [first, first+count)
            first = ref & kBvhLeafFirst;
            count = (ref >> kBvhLeafCountShift) & 15u;
            ref = pop_ref(s_stack, sp, tid);
        }
#if RTX_DET
        for (uint32_t k = 0; k < count; ++k) {
            const Group G = sc.bvh_items[first + k];
            V3 ro = o, rd = d;
            enter_group<false>(sc, G.step_begin, G.step_end, ro, rd);
            const RayAux aux = make_aux(ro, rd);
            for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) {
                const Prim P = sc.prims[p];
                float t;
                const bool is_rect = prim_kind(P) != RTX_PRIM_SPHERE;
                // candidate t of this primitive alone; the window is inclusive at
                // `closest` so that ties reach the explicit rule below
                const bool ok = prim_test(P, ro, rd, aux, t0, is_rect ? closest : t1, t) && t <= closest;
                if (ok) {
                    const bool take = t < closest || best == kNone || tie_wins(p, is_rect, best, best_is_rect);
                    if (take) {
                        closest = t;
                        best = p;
                        best_is_rect = is_rect;
                    }
                }
            }
        }
#else
        for (uint32_t k = 0; k < count; ++k) fast_leaf_item(sc, sc.bvh_fast[first + k], o, d, inv, noi, t0, closest, best);
#endif
    }
    (void)best_is_rect;
    t_hit = closest;
    prim_hit = best;
    return best != kNone;
}

#if !RTX_DET
// Whether a ray can meet anything at all: the slab test against the bounds of the whole scene
// (conservative: NaNs and rays parallel to a slab count as "can").
RTX_DEV bool ray_meets_scene(const SceneView& sc, V3 o, V3 d, float t0) {
    constexpr float kBig = 1.0e18f;
    const V3 inv = v3(__builtin_amdgcn_fmed3f(frcp(d.x), -kBig, kBig), __builtin_amdgcn_fmed3f(frcp(d.y), -kBig, kBig),
                      __builtin_amdgcn_fmed3f(frcp(d.z), -kBig, kBig));
    const float lx = (sc.bounds_lo[0] - o.x) * inv.x, hx = (sc.bounds_hi[0] - o.x) * inv.x;
    const float ly = (sc.bounds_lo[1] - o.y) * inv.y, hy = (sc.bounds_hi[1] - o.y) * inv.y;
    const float lz = (sc.bounds_lo[2] - o.z) * inv.z, hz = (sc.bounds_hi[2] - o.z) * inv.z;
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(lx, hx), __builtin_fminf(ly, hy)),
                                     __builtin_fmaxf(__builtin_fminf(lz, hz), t0));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(lx, hx), __builtin_fmaxf(ly, hy)), __builtin_fmaxf(lz, hz));
    return !(tn > tf);
}
#endif

// The nearest-hit query proper (winner and distance) and, apart from it, the completion of the
// winner's record: callers that branch on the outcome complete the record inside the branch that
// uses it, so that it is not live on the path of a miss.
template <bool kBvh>
RTX_DEV bool scene_query(const SceneView& sc, V3 o, V3 d, float t0, float t1, float& t_hit, uint32_t& prim_hit) {
    if (kBvh) return query_bvh(sc, o, d, t0, t1, t_hit, prim_hit);
    return query_linear(sc, o, d, t0, t1, t_hit, prim_hit);
}
template <bool kBvh>
RTX_DEV void scene_finish(const SceneView& sc, uint32_t prim_hit, float t_hit, V3 o, V3 d, bool want_uv, HitRec& rec) {
#if RTX_DET
    finish_hit(sc, prim_hit, t_hit, o, d, want_uv, rec);
#else
    finish_scan_hit<kBvh>(sc, prim_hit, t_hit, o, d, want_uv, rec);
#endif
}
template <bool kBvh>
RTX_DEV bool scene_hit(const SceneView& sc, V3 o, V3 d, float t0, float t1, HitRec& rec, bool want_uv = false) {
    float t_hit;
    uint32_t prim_hit;
    if (!scene_query<kBvh>(sc, o, d, t0, t1, t_hit, prim_hit)) return false;
    scene_finish<kBvh>(sc, prim_hit, t_hit, o, d, want_uv, rec);
    return true;
}

// ---------------------------------------------------------------------------
// textures (Texture.h:5): ColorTexture.h:11-13, CheckerTexture.cpp:3-11, ImageTexture.h:16-29
// ---------------------------------------------------------------------------
RTX_DEV V3 image_value(const SceneView& sc, const Texture& T, float u, float v) {
    const int w = (int)T.odd, h = (int)T.even;
    int i = (int)(u * (float)w);  // ImageTexture.h:17
#if RTX_DET
    int j = (int)((double)((1.0f - v) * (float)h) - 0.001);  // ImageTexture.h:18 (the literal is a double)
#else
    int j = (int)((1.0f - v) * (float)h - 0.001f);
#endif
    i = i < 0 ? 0 : (i >= w ? w - 1 : i);  // ImageTexture.h:23-24
    j = j < 0 ? 0 : (j >= h ? h - 1 : j);
    const uint8_t* t = sc.texels + T.offset + 3u * (uint32_t)i + 3u * (uint32_t)w * (uint32_t)j;
#if RTX_DET
    return v3((float)((double)t[0] / 255.0), (float)((double)t[1] / 255.0), (float)((double)t[2] / 255.0));
#else
    return v3((float)t[0], (float)t[1], (float)t[2]) * (1.0f / 255.0f);
#endif
}

RTX_DEV V3 texture_value(const SceneView& sc, uint32_t tex, float u, float v, V3 p) {
#pragma unroll 1
    for (int level = 0; level < 16; ++level) {  // nesting depth of CheckerTexture sub-textures
        const Texture T = sc.textures[tex];
        if (T.kind == RTX_TEX_COLOR) return v3(T.r, T.g, T.b);
        if (T.kind == RTX_TEX_IMAGE) return image_value(sc, T, u, v);
        if (T.kind == RTX_TEX_SKY) {  // Scene::backgroundSky (Scene.h:24-28) of p - centre
            const V3 w = normalize(p - v3(T.r, T.g, T.b));
            const float t = 0.5f * (w.y + 1.0f);
            const V3 white = v3(1.0f, 1.0f, 1.0f);
            return white + (v3(0.5f, 0.7f, 1.0f) - white) * t;  // lerp, vec_aos.h:92-95
        }
        float sines = (fsin(T.r * p.x) * fsin(T.r * p.y)) * fsin(T.r * p.z);
        tex = sines < 0.0f ? T.odd : T.even;
    }
    return v3(0.0f, 0.0f, 0.0f);
}

// Texture::value of the hit's material texture: plain colours come with the hit record.
RTX_DEV V3 surface_color(const SceneView& sc, const HitRec& rec) {
    if (scene_has(kFeatTextured) && (rec.bits & kInfoTextured)) return texture_value(sc, rec.texture, rec.u, rec.v, rec.p);
    return rec.color;
}

// ---------------------------------------------------------------------------
// orthonormal basis, ONB.h:24-38
// ---------------------------------------------------------------------------
struct Onb {
    V3 u, v, w;
};
RTX_DEV Onb onb_from_w(V3 n) {
    Onb b;
    b.w = normalize(n);
#if RTX_DET
    V3 a = (__builtin_fabsf(b.w.x) > 0.9f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
    b.v = normalize(cross(b.w, a));
    b.u = cross(b.w, b.v);
#else
    // Any orthonormal pair around w serves: the densities sampled in this basis (cosine lobe,
    // cone towards a sphere) are symmetric about w.  Branch-free construction without the
    // second normalisation (Duff et al., "Building an Orthonormal Basis, Revisited").
    const float sign = __builtin_copysignf(1.0f, b.w.z);
    const float a = -frcp(sign + b.w.z);
    const float c = b.w.x * b.w.y * a;
    b.u = v3(1.0f + sign * b.w.x * b.w.x * a, sign * c, -sign * b.w.x);
    b.v = v3(c, sign + b.w.y * b.w.y * a, -b.w.y);
#endif
    return b;
}
RTX_DEV V3 onb_local(const Onb& b, V3 a) {  // ONB.h:24-26
    return (b.u * a.x + b.v * a.y) + b.w * a.z;
}

// ---------------------------------------------------------------------------
// light sampling: ShapeList / Rect / Sphere :: random and :: pdf_value
// ---------------------------------------------------------------------------

// MixturePdf::generate (MixturePdf.h:15-22) for a diffuse hit: one draw picks between
//   ShapePdf::generate -> Shape::random of the light shape (ShapePdf.h:15-17; for a list one
//     more draw picks the entry, ShapeList.cpp:28-35; then Rect.cpp:50-68 or Sphere.cpp:52-58
//     with random_to_sphere, Util.h:96-107; any other shape returns (1,0,0), Shape.h:9), and
//   CosinePdf::generate (CosinePdf.cpp:16-20) with random_cosine_direction (Util.h:86-94).
// The three samplers that draw (XZ rect, sphere, cosine) all consume exactly two more
// numbers, and the sphere and cosine samplers share their shape: an orthonormal basis around
// an axis, a polar angle from the second number, the azimuth 2*pi*r1 from the first.  They
// are therefore evaluated by one code path; each lane still performs exactly the operations
// of its own sampler, in the reference's order.
constexpr uint32_t kSampleCosine = 5u;  // beyond the RTX_LIGHT_* kinds

// cos(theta_max) of the cone a sphere fills seen from a point: sqrt(1 - rr / dd) with rr =
// min(r^2, dd) (Sphere.cpp:44-46, 55-56).  From inside the sphere rr == dd; the approximate
// reciprocal of fast mode can then land a hair above 1: clamp, so that the root is 0 and not NaN.
RTX_DEV float cone_cosine(float rr, float dd) {
#if RTX_DET
    return fsqrt(1.0f - rr * frcp(dd));
#else
    // sqrt((dd - rr) / dd) with one reciprocal square root: x * rsqrt(x * dd); x = 0 (inside the
    // sphere) gives 0, not NaN, through the floor under the product
    const float x = __builtin_fmaxf(dd - rr, 0.0f);
    return x * finvsqrt(__builtin_fmaxf(x * dd, 1.0e-30f));
#endif
}

// kListLit (here and below): the caller knows that the light shape is a list with at least one entry
// and that the background is a constant colour - what the reference's own scene is (Scene.cpp:83-86,
// Scene.cpp:131).  The wave-uniform tests for it then need no registers across the kernel's loop.
template <bool kListLit = false>
RTX_DEV V3 sample_direction(const SceneView& sc, V3 o, V3 n, RtxRng& rng) {
    uint32_t kind = kSampleCosine;
    Light E;
    E.a = E.b = E.c = E.d = E.e = 0.0f;
    // no light shape: the material's own pdf alone, without the mixture's draw (wave-uniform)
    if ((kListLit || sc.n_lights != 0u) && rng_nextf(&rng) < 0.5f) {
        uint32_t index = 0;
        if (kListLit || sc.light_is_list) {
            float r = rng_nextf(&rng);
#if RTX_DET
            index = min((uint32_t)(r * (float)sc.n_lights), sc.n_lights - 1u);
#else
            index = min((uint32_t)(r * sc.n_lights_f), sc.n_lights - 1u);
#endif
        }
        E = at32(sc.lights, index);  // per-lane
        kind = E.kind;
    }
    const bool is_sphere = scene_has(kFeatLightSphere) && kind == RTX_LIGHT_SPHERE;
    const bool is_rect = scene_has(kFeatLightRect) && kind == RTX_LIGHT_RECT_XZ;
    if (scene_has(kFeatLightOther) && !(is_sphere | is_rect | (kind == kSampleCosine))) return v3(1.0f, 0.0f, 0.0f);
    const float r1 = rng_nextf(&rng);
    const float r2 = rng_nextf(&rng);
    if (is_rect) {
        float x = E.a + r1 * (E.b - E.a);
        float y = E.c + r2 * (E.d - E.c);
        return v3(x, E.e, y) - o;
    }
    const V3 axis = is_sphere ? v3(E.a, E.b, E.c) - o : n;
    const Onb uvw = onb_from_w(axis);
    float z, radial;
    if (is_sphere) {
        float dd = length_sqr(axis);
        float r2r = E.d * E.d;
        float rr = dd < r2r ? dd : r2r;  // std::min(pow2(radius), distance_squared)
        float cos_theta_max = cone_cosine(rr, dd);
        z = 1.0f - r2 * (1.0f - cos_theta_max);
        radial = fsqrt(1.0f - z * z);
    } else {
        z = fsqrt(1.0f - r2);
        radial = fsqrt(r2);
    }
    float sn, cs;
    sincos_2pi(r1, sn, cs);
    return onb_local(uvw, v3(cs * radial, sn * radial, z));
}

// Shape::pdf_value of one light entry (Rect.cpp:36-48, Sphere.cpp:38-50, Shape.h:8)
RTX_DEV float light_pdf_one(const Light& E, V3 o, V3 v) {
    if (E.kind == RTX_LIGHT_RECT_XZ) {
        float t;
        const RayAux aux = make_aux(o, v);
        if (!rect_axis<0, 2, 1>(E.a, E.b, E.c, E.d, E.e, o, v, aux, kTMin, kFltMax, t)) return 0.0f;
        float area = (E.b - E.a) * (E.d - E.c);
        float distance_squared = (t * t) * length_sqr(v);
        float cosine = fdiv(__builtin_fabsf(v.y), length(v));  // |dot(v, (0,1,0))| / |v|
        return fdiv(distance_squared, cosine * area);
    }
    if (E.kind == RTX_LIGHT_SPHERE) {
        float t;
        if (!sphere_test(E.a, E.b, E.c, E.d, o, v, kTMin, kFltMax, t)) return 0.0f;
        float dd = length_sqr(v3(E.a, E.b, E.c) - o);
        float r2r = E.d * E.d;
        float rr = dd < r2r ? dd : r2r;
        float cos_theta_max = cone_cosine(rr, dd);
        float solid_angle = kPi2 * (1.0f - cos_theta_max);
        return frcp(solid_angle);
    }
    return 0.0f;
}

// ShapePdf::value (ShapePdf.h:11-13); list: ShapeList.cpp:19-26
RTX_DEV float light_pdf(const SceneView& sc, V3 o, V3 v) {
    if (!sc.light_is_list) {
        const Light E = load_uniform(sc.lights, 0u);
        return light_pdf_one(E, o, v);
    }
    float weight = fdiv(1.0f, (float)sc.n_lights);
    float sum = 0.0f;
    for (uint32_t k = 0; k < sc.n_lights; ++k) {
        const Light E = load_uniform(sc.lights, k);
        sum = sum + weight * light_pdf_one(E, o, v);
    }
    return sum;
}

#if !RTX_DET
// The same densities with the algebra folded (fast mode): len2 = |v|^2 and inv_len = 1/|v| come
// from the caller, which needs them for the cosine density too.
//   rect (Rect.cpp:36-48): t^2 |v|^2 / (|v.y| / |v| * area) with t = (k - o.y) / v.y
//   sphere (Sphere.cpp:38-50): hit iff the far root lies beyond t_min; 1 / (2 pi (1 - cos_max))
RTX_DEV float light_pdf_one_fast(const Light& E, V3 o, V3 v, float len2, float inv_len) {
    if (E.kind == RTX_LIGHT_RECT_XZ) {
        const float inv_vy = frcp(v.y);
        const float t = (E.e - o.y) * inv_vy;
        const float x = __builtin_fmaf(t, v.x, o.x), z = __builtin_fmaf(t, v.z, o.z);
        const bool ok = (t >= kTMin) & (x >= E.a) & (x <= E.b) & (z >= E.c) & (z <= E.d);
        const float pdf = ((t * t) * (len2 * len2)) * ((inv_len * __builtin_fabsf(inv_vy)) * E.inv_area);
        return ok ? pdf : 0.0f;
    }
    if (E.kind == RTX_LIGHT_SPHERE) {
        const V3 oc = o - v3(E.a, E.b, E.c);
        const float h = dot(oc, v), dd = dot(oc, oc), r2 = E.d * E.d;
        const float D = h * h - len2 * (dd - r2);
        const bool ok = (D > 0.0f) & (fsqrt(D) - h > kTMin * len2);
        const float rr = dd < r2 ? dd : r2;
        const float cos_theta_max = cone_cosine(rr, dd);
        return ok ? frcp(kPi2 * (1.0f - cos_theta_max)) : 0.0f;
    }
    return 0.0f;
}

template <bool kListLit = false>
RTX_DEV float light_pdf_fast(const SceneView& sc, V3 o, V3 v, float len2, float inv_len) {
#ifdef RTX_SPECIAL_LIGHTS
    // the light entries of the scene this kernel was compiled for, as literals (rtx_special.cpp): the
    // sum below in straight-line code, each entry's kind resolved at compile time
    {
        float sum_ = 0.0f;
#define RTX_SPECIAL_LIGHT(KIND, A, B, C, D, E, INV_AREA)                                                             \
    do {                                                                                                             \
        Light E_;                                                                                                    \
        E_.a = RTX_SPECIAL_F(A); E_.b = RTX_SPECIAL_F(B); E_.c = RTX_SPECIAL_F(C); E_.d = RTX_SPECIAL_F(D);          \
        E_.e = RTX_SPECIAL_F(E); E_.kind = (KIND); E_.inv_area = RTX_SPECIAL_F(INV_AREA); E_.pad = 0u;               \
        sum_ += light_pdf_one_fast(E_, o, v, len2, inv_len);                                                         \
    } while (0)
        RTX_SPECIAL_LIGHTS
#undef RTX_SPECIAL_LIGHT
        return sum_ * RTX_SPECIAL_F(RTX_SPECIAL_LIGHT_WEIGHT);
    }
#endif
    if (!kListLit && !sc.light_is_list) return light_pdf_one_fast(load_uniform(sc.lights, 0u), o, v, len2, inv_len);
    float sum = 0.0f;
    if (kListLit) {  // at least one entry: no test before the first turn (and no register for its outcome)
        uint32_t k = 0u;
        do sum += light_pdf_one_fast(load_uniform(sc.lights, k), o, v, len2, inv_len);
        while (++k < sc.n_lights);
    } else {
        for (uint32_t k = 0; k < sc.n_lights; ++k) sum += light_pdf_one_fast(load_uniform(sc.lights, k), o, v, len2, inv_len);
    }
    return sum * sc.inv_n_lights;
}
#endif


// ---------------------------------------------------------------------------
// light trees and general light shapes (kernels instantiated with kLightTree)
// ---------------------------------------------------------------------------
// The light shape as a tree: ShapeList entries may be ShapeLists again (ShapeList.cpp:19-35
// recurse: random() draws one more number per level, pdf_value() weighs each level by 1 / n),
// and - with general light sampling, which the reference does not have - a leaf may be a rect of
// any axis and may sit under Translate / Rotate wrappers.  A wrapped leaf answers in its own
// frame: the query point (and direction) enter it as a ray enters Translate::hit / Rotate::hit
// (Translate.cpp:7, Rotate.cpp:7-9) and a sampled direction leaves it as a normal does
// (Rotate.cpp:13; a translation does not change a direction).

// Shape::pdf_value of a leaf for a query already in the leaf's frame: Rect.cpp:36-48 (for the
// rect's own axis instead of XZ only when `general`), Sphere.cpp:38-50, Shape.h:8.
RTX_DEV float light_pdf_leaf(const Light& E, V3 o, V3 v, bool general) {
    const uint32_t kind = E.kind;
    if (kind == RTX_LIGHT_RECT_XZ || (general && kind <= RTX_LIGHT_RECT_YZ)) {
        float t;
        const RayAux aux = make_aux(o, v);
        if (!rect_test(kind, E.a, E.b, E.c, E.d, E.e, o, v, aux, kTMin, kFltMax, t)) return 0.0f;
        const float vn = kind == RTX_LIGHT_RECT_XY ? v.z : (kind == RTX_LIGHT_RECT_XZ ? v.y : v.x);
        float area = (E.b - E.a) * (E.d - E.c);
        float distance_squared = (t * t) * length_sqr(v);
        float cosine = fdiv(__builtin_fabsf(vn), length(v));
        return fdiv(distance_squared, cosine * area);
    }
    if (kind == RTX_LIGHT_SPHERE) {
        float t;
        if (!sphere_test(E.a, E.b, E.c, E.d, o, v, kTMin, kFltMax, t)) return 0.0f;
        float dd = length_sqr(v3(E.a, E.b, E.c) - o);
        float r2r = E.d * E.d;
        float rr = dd < r2r ? dd : r2r;
        float cos_theta_max = cone_cosine(rr, dd);
        float solid_angle = kPi2 * (1.0f - cos_theta_max);
        return frcp(solid_angle);
    }
    return 0.0f;
}

// ShapePdf::value over the whole tree: the sum over the leaves of (product of the 1 / n above
// the leaf) * its density.  The reference nests the sums (ShapeList.cpp:19-26); for a flat list
// this is the same sum in the same order, for nested lists it differs by rounding only.
RTX_DEV float light_pdf_tree(const SceneView& sc, V3 o, V3 v) {
    const bool general = (sc.light_flags & kLightsGeneral) != 0u;
    float sum = 0.0f;
    if (sc.n_light_bvh != 0u) {
        // Many lights: only the leaves whose box the ray (o, v), t >= t_min, meets can answer
        // anything but 0.  Per lane, stackless: depth-first order with skip links.  The terms are
        // added in the tree's order, not the list's: the same sum up to rounding.
        constexpr float kBig = 1.0e18f;  // a direction component of zero: keep the reciprocal finite
        const V3 inv = v3(__builtin_amdgcn_fmed3f(frcp(v.x), -kBig, kBig), __builtin_amdgcn_fmed3f(frcp(v.y), -kBig, kBig),
                          __builtin_amdgcn_fmed3f(frcp(v.z), -kBig, kBig));
        const V3 noi = v3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
        uint32_t i = 0u;
        while (i < sc.n_light_bvh) {
            const LightBvhNode B = sc.light_bvh[i];
            const float tx0 = __builtin_fmaf(B.lo[0], inv.x, noi.x), tx1 = __builtin_fmaf(B.hi[0], inv.x, noi.x);
            const float ty0 = __builtin_fmaf(B.lo[1], inv.y, noi.y), ty1 = __builtin_fmaf(B.hi[1], inv.y, noi.y);
            const float tz0 = __builtin_fmaf(B.lo[2], inv.z, noi.z), tz1 = __builtin_fmaf(B.hi[2], inv.z, noi.z);
            const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.0f));
            const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
            if (!(tn <= tf)) {
                i = B.skip;
                continue;
            }
            if (B.entry != kLightBvhInner) {
                const Light E = sc.lights[B.entry];
                const LightNode N = sc.light_nodes[B.entry];
                V3 ol = o, vl = v;
                enter_group<false>(sc, N.step_begin, N.step_end, ol, vl);
                sum = sum + N.weight * light_pdf_leaf(E, ol, vl, general);
            }
            ++i;  // the first child, or the node after a leaf (= its skip link)
        }
        return sum;
    }
    // consecutive leaves under the same chain (the six faces of a box) share the transformed query
    uint32_t cur_begin = 0u, cur_end = 0u;
    V3 ol = o, vl = v;
    for (uint32_t e = 0; e < sc.n_light_entries; ++e) {
        const Light E = load_uniform(sc.lights, e);
        if (E.kind >= RTX_LIGHT_OTHER) continue;  // no density of its own (Shape.h:8) / a list
        const LightNode N = load_uniform(sc.light_nodes, e);
        if (N.step_begin != cur_begin || N.step_end != cur_end) {  // wave-uniform
            cur_begin = N.step_begin;
            cur_end = N.step_end;
            ol = o;
            vl = v;
            enter_group<true>(sc, cur_begin, cur_end, ol, vl);
        }
        sum = sum + N.weight * light_pdf_leaf(E, ol, vl, general);
    }
    return sum;
}

// MixturePdf::generate (MixturePdf.h:15-22) with the light shape as a tree.
RTX_DEV V3 sample_direction_tree(const SceneView& sc, V3 o, V3 n, RtxRng& rng) {
    if (sc.n_lights != 0u && rng_nextf(&rng) < 0.5f) {
        uint32_t index = 0;
        if (sc.light_is_list) {  // ShapeList::random, ShapeList.cpp:28-35
            float r = rng_nextf(&rng);
            index = (uint32_t)(r * (float)sc.n_lights);
            if (index >= sc.n_lights) index = sc.n_lights - 1;
        }
        Light E = sc.lights[index];  // per lane
        for (int level = 0; level < 64 && E.kind == RTX_LIGHT_LIST; ++level) {  // a nested list picks again (the upload bounds the nesting)
            const LightNode L = sc.light_nodes[index];
            float r = rng_nextf(&rng);
            uint32_t k = (uint32_t)(r * (float)L.count);
            if (k >= L.count) k = L.count - 1;
            index = L.first + k;
            E = sc.lights[index];
        }
        const bool general = (sc.light_flags & kLightsGeneral) != 0u;
        const bool is_rect = E.kind == RTX_LIGHT_RECT_XZ || (general && E.kind <= RTX_LIGHT_RECT_YZ);
        if (!is_rect && E.kind != RTX_LIGHT_SPHERE) return v3(1.0f, 0.0f, 0.0f);  // Shape.h:9
        const LightNode N = sc.light_nodes[index];
        V3 ol = o, unused = v3(0.0f, 0.0f, 0.0f);
        enter_group<false>(sc, N.step_begin, N.step_end, ol, unused);
        V3 dir;
        if (is_rect) {  // Rect.cpp:50-68, for the rect's own axis
            float x = E.a + rng_nextf(&rng) * (E.b - E.a);
            float y = E.c + rng_nextf(&rng) * (E.d - E.c);
            const V3 point = E.kind == RTX_LIGHT_RECT_XY ? v3(x, y, E.e)
                                                         : (E.kind == RTX_LIGHT_RECT_XZ ? v3(x, E.e, y) : v3(E.e, x, y));
            dir = point - ol;
        } else {  // Sphere.cpp:52-58 with random_to_sphere, Util.h:96-107
            const V3 axis = v3(E.a, E.b, E.c) - ol;
            float dd = length_sqr(axis);
            const Onb uvw = onb_from_w(axis);
            const float r1 = rng_nextf(&rng);
            const float r2 = rng_nextf(&rng);
            float r2r = E.d * E.d;
            float rr = dd < r2r ? dd : r2r;
            float cos_theta_max = cone_cosine(rr, dd);
            float z = 1.0f - r2 * (1.0f - cos_theta_max);
            float radial = fsqrt(1.0f - z * z);
            float sn, cs;
            sincos_2pi(r1, sn, cs);
            dir = onb_local(uvw, v3(cs * radial, sn * radial, z));
        }
        for (uint32_t s = N.step_end; s > N.step_begin; --s) {  // out of the leaf's frame, innermost wrapper first
            const Step S = sc.steps[s - 1];
            if (S.kind == RTX_STEP_ROTATE) dir = qrotate(S.x, S.y, S.z, S.w, dir);
        }
        return dir;
    }
    // CosinePdf::generate (CosinePdf.cpp:16-20) with random_cosine_direction (Util.h:86-94)
    const Onb uvw = onb_from_w(n);
    const float r1 = rng_nextf(&rng);
    const float r2 = rng_nextf(&rng);
    float z = fsqrt(1.0f - r2);
    float radial = fsqrt(r2);
    float sn, cs;
    sincos_2pi(r1, sn, cs);
    return onb_local(uvw, v3(cs * radial, sn * radial, z));
}

// ---------------------------------------------------------------------------
// reflect / refract / schlick, Util.h:56-76
// ---------------------------------------------------------------------------
RTX_DEV V3 reflect(V3 v, V3 n) { return v - n * (2.0f * dot(v, n)); }

RTX_DEV bool refract(V3 v, V3 n, float ni_over_nt, V3& refracted) {
    V3 uv = normalize(v);
    float dt = dot(uv, n);
    float D = 1.0f - (ni_over_nt * ni_over_nt) * (1.0f - dt * dt);
    if (D > 0.0f) {
        refracted = (uv - n * dt) * (-ni_over_nt) - n * fsqrt(D);
        return true;
    }
    return false;
}

RTX_DEV float schlick(float cosine, float ri) {
    float q = fdiv(1.0f - ri, 1.0f + ri);
    float r0 = q * q;
    float m = 1.0f - cosine;
    float m5 = (((m * m) * m) * m) * m;  // pow5, Util.h:17
    return r0 + (1.0f - r0) * m5;
}

// ---------------------------------------------------------------------------
// one path vertex: Scene::color (Scene.cpp:103-132) unrolled into a loop body.
// Returns true when the path ends; `L` is then the sample's radiance.
// ---------------------------------------------------------------------------
struct PathState {
    V3 o, d;       // current ray
    V3 T;          // product of the per-vertex weights so far
    uint32_t depth;
    RtxRng rng;
};

// On the branches that end a path the state is left alone.  The empty statement below redefines
// every field IN PLACE (operands tied to themselves), which keeps the register allocator from
// giving the state after shade() other registers than the state before it - and from copying the
// one into the other on every path-ending branch and back at the end of the loop.
RTX_DEV bool pin(PathState& ps) {
    asm("" : "+v"(ps.o.x), "+v"(ps.o.y), "+v"(ps.o.z), "+v"(ps.d.x), "+v"(ps.d.y), "+v"(ps.d.z), "+v"(ps.T.x), "+v"(ps.T.y),
             "+v"(ps.T.z), "+v"(ps.depth), "+v"(ps.rng.state));
    return true;
}

template <bool kLightTree, bool kListLit = false>
RTX_DEV bool shade(const SceneView& sc, uint32_t max_depth, bool hit, const HitRec& rec, PathState& ps,
                   V3& L) {
    if (!hit) {  // Scene.cpp:131
        V3 bg = v3(sc.background);
        if (!kListLit && sc.background_mode == RTX_BACKGROUND_SKY) {  // Scene::backgroundSky, Scene.h:24-28
            const V3 v = normalize(ps.d);
            const float t = 0.5f * (v.y + 1.0f);
            const V3 white = v3(1.0f, 1.0f, 1.0f);
            bg = white + (v3(0.5f, 0.7f, 1.0f) - white) * t;  // lerp, vec_aos.h:92-95
        }
        L = vmul(ps.T, bg);
        return pin(ps);
    }
    const uint32_t mat_kind = rec_mat_kind(rec);
    if (scene_has(kFeatEmitter) && mat_kind == RTX_MAT_DIFFUSE_LIGHT) {  // DiffuseLight.h:17-28: emits on the front face, never scatters
        V3 e = v3(0.0f, 0.0f, 0.0f);
        if (dot(rec.n, ps.d) < 0.0f) e = surface_color(sc, rec);
        L = vmul(ps.T, e);
        return pin(ps);
    }
    // the other materials emit nothing (Material.h:10)
    L = v3(0.0f, 0.0f, 0.0f);
    if (ps.depth >= max_depth) return pin(ps);  // Scene.cpp:108

    // (a scene without Metal and Dielectric: what is not an emitter is a Lambertian)
    if (scene_has(kFeatLambertian) && (!scene_has(kFeatMetal | kFeatDielectric) || mat_kind == RTX_MAT_LAMBERTIAN)) {
        V3 albedo = surface_color(sc, rec);  // Lambertian.cpp:17
        const V3 dir = kLightTree ? sample_direction_tree(sc, rec.p, rec.n, ps.rng)
                                  : sample_direction<kListLit>(sc, rec.p, rec.n, ps.rng);
#if !RTX_DET
        if (!kLightTree) {  // the same estimator with 1 / |dir| shared between the densities
            const float len2 = dot(dir, dir), inv_len = finvsqrt(len2);
            const float cosine = dot(dir, rec.n) * inv_len;
            const float pdf1 = (cosine > 0.0f ? cosine : 0.0f) * (1.0f / kPi);  // = the scattering pdf
            const float pdf_value =
                (kListLit || sc.n_lights != 0u) ? 0.5f * light_pdf_fast<kListLit>(sc, rec.p, dir, len2, inv_len) + 0.5f * pdf1 : pdf1;
            // the next ray is committed before the test: a path that ends here has no use for the
            // old one either, and the old one's registers are free from this point on
            ps.o = rec.p;
            ps.d = dir;
            ps.depth += 1;
            if (!(pdf_value > 0.0f)) return pin(ps);
            ps.T = vmul(albedo, ps.T) * (pdf1 * frcp(pdf_value));
            return false;
        }
#endif
        // MixturePdf::value, MixturePdf.h:9-13
        float cosine = dot(normalize(dir), rec.n);  // CosinePdf.cpp:7
        float pdf1 = cosine > 0.0f ? fdiv(cosine, kPi) : 0.0f;
        float pdf_value = pdf1;  // no light shape: srec.pdf alone
        if (sc.n_lights != 0u)
            pdf_value = 0.5f * (kLightTree ? light_pdf_tree(sc, rec.p, dir) : light_pdf(sc, rec.p, dir)) + 0.5f * pdf1;
        if (!(pdf_value > 0.0f)) return pin(ps);  // Scene.cpp:117,122-124
        // Lambertian::scattering_pdf, Lambertian.cpp:23-25: std::max(cos, 0) / PI
        float c0 = cosine < 0.0f ? 0.0f : cosine;
        float spdf = fdiv(c0, kPi);
        ps.T = vdivs(vmul(albedo * spdf, ps.T), pdf_value);  // Scene.cpp:119-120
        ps.o = rec.p;
        ps.d = dir;
        ps.depth += 1;
        return false;
    }
#if !RTX_DET
    {   // Metal (Metal.cpp:9-17) and Dielectric (Dielectric.cpp:7-45) in one pass: both mirror the
        // incoming direction about the normal - Metal the normalised one, Dielectric the raw one -
        // so the few lanes of either kind share that part instead of running two separate blocks.
        const bool metal = scene_has(kFeatMetal) && (!scene_has(kFeatDielectric) || mat_kind == RTX_MAT_METAL);
        const float dn = dot(ps.d, rec.n);
        const float inv_len = finvsqrt(dot(ps.d, ps.d));
        const float s = metal ? inv_len : 1.0f;
        V3 reflected = ps.d * s - rec.n * ((2.0f * dn) * s);  // reflect, Util.h:56-58
        V3 dir = reflected;
        if (metal) {
            if (scene_has(kFeatFuzz) && rec.param != 0.0f) {  // a perfect mirror ignores the fuzz vector
                V3 q;
                do {  // random_in_unit_sphere, Util.h:32-38
                    float rz = rng_nextf(&ps.rng);
                    float ry = rng_nextf(&ps.rng);
                    float rx = rng_nextf(&ps.rng);
                    q = v3(rx, ry, rz) * 2.0f - v3(1.0f, 1.0f, 1.0f);
                } while (length_sqr(q) >= 1.0f);
                reflected = reflected + q * rec.param;
                dir = reflected;
            }
        } else {
            const float ri = rec.param;
            const bool inside = dn > 0.0f;
            const V3 outward = inside ? -rec.n : rec.n;
            // (the hit table holds 1 / ri and Schlick's r0 in the colour slots of a Dielectric)
            const float ni_over_nt = inside ? ri : rec.color.x;
            const float dt = __builtin_fabsf(dn) * inv_len;          // dot(normalize(-d), outward)
            const float cosine = inside ? ri * dt : dt;               // Dielectric.cpp:19,24
            const float D = 1.0f - (ni_over_nt * ni_over_nt) * (1.0f - dt * dt);  // refract, Util.h:60-71
            float reflect_prob = 1.0f;
            V3 refracted = v3(0.0f, 0.0f, 0.0f);
            if (D > 0.0f) {
                const V3 uv = ps.d * (-inv_len);
                refracted = (uv - outward * dt) * (-ni_over_nt) - outward * fsqrt(D);
                const float m = 1.0f - cosine, m2 = m * m, r0 = rec.color.y;
                reflect_prob = r0 + (1.0f - r0) * ((m2 * m2) * m);  // schlick, Util.h:73-76
            }
            if (!(rng_nextf(&ps.rng) < reflect_prob)) dir = refracted;
            // albedo is (1,1,1): the weight is unchanged
        }
        ps.o = rec.p;
        ps.d = dir;
        ps.depth += 1;
        if (metal) {  // (tested after the commit of the next ray, as above)
            if (!(dot(dir, rec.n) > 0.0f)) return pin(ps);
            ps.T = vmul(surface_color(sc, rec), ps.T);  // Scene.cpp:110
        }
        return false;
    }
#endif
    if (mat_kind == RTX_MAT_METAL) {  // Metal.cpp:9-17
        V3 reflected = reflect(normalize(ps.d), rec.n);
#if !RTX_DET
        // a perfect mirror ignores the fuzz vector; only the deterministic mode has to
        // consume the reference's draws for it
        if (rec.param != 0.0f)
#endif
        {
            V3 q;
            do {  // random_in_unit_sphere, Util.h:32-38; runs even when fuzz == 0.
                // The oracle build (g++) evaluates the three draws of random_vector()
                // (Util.h:28-30) right to left: z, y, x.
                float rz = rng_nextf(&ps.rng);
                float ry = rng_nextf(&ps.rng);
                float rx = rng_nextf(&ps.rng);
                q = v3(rx, ry, rz) * 2.0f - v3(1.0f, 1.0f, 1.0f);
            } while (length_sqr(q) >= 1.0f);
            reflected = reflected + q * rec.param;
        }
        V3 albedo = surface_color(sc, rec);
        if (!(dot(reflected, rec.n) > 0.0f)) return pin(ps);
        ps.T = vmul(albedo, ps.T);  // Scene.cpp:110
        ps.o = rec.p;
        ps.d = reflected;
        ps.depth += 1;
        return false;
    }
    // Dielectric, Dielectric.cpp:7-45
    {
        float ri = rec.param;
        V3 reflected = reflect(ps.d, rec.n);
        V3 outward;
        float ni_over_nt, cosine;
        float dn = dot(ps.d, rec.n);
        if (dn > 0.0f) {
            outward = -rec.n;
            ni_over_nt = ri;
            cosine = fdiv(ri * dn, length(ps.d));
        } else {
            outward = rec.n;
            ni_over_nt = frcp(ri);
            cosine = fdiv(-dn, length(ps.d));
        }
        V3 refracted = v3(0.0f, 0.0f, 0.0f);
        float reflect_prob = 1.0f;
        if (refract(-ps.d, outward, ni_over_nt, refracted)) reflect_prob = schlick(cosine, ri);
        V3 dir = (rng_nextf(&ps.rng) < reflect_prob) ? reflected : refracted;
        // albedo is (1,1,1): the weight is unchanged (mulPerElem by one is exact)
        ps.o = rec.p;
        ps.d = dir;
        ps.depth += 1;
        return false;
    }
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

// The loops of Scene::render (Scene.cpp:140-149) around the integrator.
//
// Work distribution.  A wave owns a tile of P = 4 .. 64 pixels (2x2 .. 8x8, chosen by the
// launcher from the sample count; 2x2 waves per 256-thread block) and the pool of its
// (pixel, sample) items: item k is pixel slot k mod P of the tile, sample
// spp_begin + k / P.  A lane whose path has ended draws the next item of the pool
// (wave64 ballot + prefix count), so every lane runs the nearest-hit query each iteration
// until the pool is empty, whatever the path lengths of the individual pixels.  A finished
// sample is added to its pixel's sum in LDS (one fp32 add per channel); lane l < P writes
// pixel slot l to HBM once, at the end.  Only this wave touches its slice of LDS, and the wave
// executes in lockstep, so the order of the adds - and with it every bit of the result -
// is reproducible from run to run.
template <bool kBvh, bool kSamples, bool kLightTree, bool kListLit = false>
// 96 scalar registers per wave: a SIMD's scalar register file holds eight waves of that size; at
// the 106 the kernel would take by itself the ninth-to-sixth waves do not fit, whatever the vector
// registers allow (measured: 83.6 -> 85.2 Grays/s on the Cornell box).
__attribute__((amdgpu_num_sgpr(RTX_TRACE_SGPRS))) __global__ void __launch_bounds__(RTX_BLOCK, kBvh ? RTX_TRACE_WAVES_BVH : RTX_TRACE_WAVES) trace_kernel(const TraceArgs a) {
    // One block of LDS per wave - the pixel sums and (fast mode) the per-slot start-up data - so that one
    // scalar register holds the base of both.
    struct alignas(16) WaveLds {
#if !RTX_DET
        uint4 slot[64];
        float4 pixel_u, pixel_v;  // one pixel's step along the two image axes (a broadcast read per new item)
#endif
        float acc[3][64];
    };
    __shared__ WaveLds s_wave[RTX_BLOCK / 64];
#ifdef RTX_PROBE_LDS_BYTES  // occupancy experiments: LDS ballast that caps the resident blocks per CU
    __shared__ float s_ballast[RTX_PROBE_LDS_BYTES / 4];
    if (a.width == 0xffffffffu) s_ballast[threadIdx.x] = 1.0f;
#endif
    const SceneView& sc = a.scene;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t twl = a.tile_w_log2, pixl = a.tile_w_log2 + a.tile_h_log2;  // this wave's pixel tile
    const uint32_t tile_w = 1u << twl, tile_h = 1u << a.tile_h_log2;
    [[maybe_unused]] const uint32_t pix_mask = (1u << pixl) - 1u;
    constexpr uint32_t kSide = RTX_BLOCK == 256 ? 2u : 1u;  // waves per block edge
    const uint32_t tiles_x = (a.width + kSide * tile_w - 1u) / (kSide * tile_w);
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t x0 = (kSide * bx + (wave & (kSide - 1u))) * tile_w;  // tile origin; rows counted from the bottom
    const uint32_t y0 = (kSide * by + (wave >> (kSide - 1u))) * tile_h;
    if (x0 >= a.width || y0 >= a.height) return;  // wave-uniform
    const uint32_t tw = min(tile_w, a.width - x0), th = min(tile_h, a.height - y0);

    s_wave[wave].acc[0][lane] = 0.0f;
    s_wave[wave].acc[1][lane] = 0.0f;
    s_wave[wave].acc[2][lane] = 0.0f;
#if !RTX_DET
    // What a new item needs of its pixel - the stream key, the coordinates as floats, whether the
    // slot lies inside the image - is the same for all the samples of the pixel: lane l works it
    // out once for slot l, and starting an item costs one 16-byte LDS read instead of a hash, an
    // integer multiply-add and two conversions.
    const float inv_nx = frcp((float)a.width), inv_ny = frcp((float)a.height);
    // one pixel's step along the two image axes (wave-uniform: back into scalar registers)
    // one pixel's step along the two image axes: kept in LDS, not in six scalar registers for the whole loop
    if (lane == 0u) {
        s_wave[wave].pixel_u = make_float4(sc.cam_u[0] * inv_nx, sc.cam_u[1] * inv_nx, sc.cam_u[2] * inv_nx, 0.0f);
        s_wave[wave].pixel_v = make_float4(sc.cam_v[0] * inv_ny, sc.cam_v[1] * inv_ny, sc.cam_v[2] * inv_ny, 0.0f);
    }
    {
        const uint32_t sx = lane & (tile_w - 1u), sy = lane >> twl;
        const uint32_t i = x0 + sx, j = y0 + sy;
        const bool inside = lane <= pix_mask && sx < tw && sy < th;
        // direction of the ray through the pixel's lower left corner (Camera::getRay, Camera.h:24-26);
        // a sample adds its jitter along the two image axes
        const float u0 = (float)i * inv_nx, v0 = (float)j * inv_ny;
        const V3 corner = ((v3(sc.cam_ll) - v3(sc.cam_origin)) + v3(sc.cam_u) * u0) + v3(sc.cam_v) * v0;
        const uint32_t key = rng_pixel_key(j * a.width + i, a.seed);
        s_wave[wave].slot[lane] = make_uint4(inside ? (key != 0u ? key : 1u) : 0u, __float_as_uint(corner.x),
                                        __float_as_uint(corner.y), __float_as_uint(corner.z));
    }
#endif

    const uint32_t total = a.spp_count << pixl;  // items of this wave's pool
    uint32_t next = 0;                         // wave-uniform pool cursor
    uint32_t rays = 0, segs = 0, item = 0;
    bool active = false;
    PathState ps;
    ps.o = ps.d = ps.T = v3(0.0f, 0.0f, 0.0f);
    ps.depth = 0;
    ps.rng.state = 0;
    ps.rng.inc = 1;
#if RTX_DET
    const float fnx = (float)a.width, fny = (float)a.height;
#endif

    for (;;) {
        const unsigned long long want = rtx_ballot(!active);
        // (all 64 lanes of the wave are here; with the pool spent nothing below can start a path.  A
        // draw that found only slots outside the image - a ragged edge tile - costs one more turn of
        // the harmless body before this test sees it.)
        if (want == ~0ull && next >= total) break;
        if (want != 0ull && next < total) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
            const uint32_t mine = next + rank;
            next += (uint32_t)__popcll(want);
            if (!active && mine < total) {
                const uint32_t slot = __builtin_amdgcn_ubfe(mine, 0u, pixl);  // mine & pix_mask from the one register that also shifts
#if RTX_DET
                const uint32_t sx = slot & (tile_w - 1u), sy = slot >> twl;
                if (sx < tw && sy < th) {
                    item = mine;
                    const uint32_t i = x0 + sx, j = y0 + sy;
                    rng_init(&ps.rng, j * a.width + i, a.spp_begin + (mine >> pixl), a.seed);
                    const float fi = (float)i, fj = (float)j;
#else
                const uint4 slot_info = s_wave[wave].slot[slot];
                if (slot_info.x != 0u) {  // 0: the slot lies outside the image
                    item = mine;
                    rng_init_keyed(&ps.rng, slot_info.x, a.spp_begin + (mine >> pixl));
                    const float du = rng_nextf(&ps.rng);  // Scene.cpp:145-146: u first, then v
                    const float dv = rng_nextf(&ps.rng);
                    ps.o = v3(sc.cam_origin);
                    // (the origin from here on as the vector registers just written, not as three scalar
                    // operands again: an instruction with a scalar operand issues at half rate)
                    asm("" : "+v"(ps.o.x), "+v"(ps.o.y), "+v"(ps.o.z));
                    const float4 pu = s_wave[wave].pixel_u, pv = s_wave[wave].pixel_v;
                    ps.d = (v3(__uint_as_float(slot_info.y), __uint_as_float(slot_info.z), __uint_as_float(slot_info.w)) +
                            v3(pu.x, pu.y, pu.z) * du) + v3(pv.x, pv.y, pv.z) * dv;
                    // The reference forms the direction as (lower-left + ...) - origin (Camera.h:24-26),
                    // i.e. it rounds the point on the image plane in WORLD coordinates before it
                    // subtracts the origin, which puts the directions on a grid of one ulp of those
                    // coordinates.  That grid is visible in the statistics (Cornell box: 0.11 % fewer
                    // rays per sample, 0.15 % more radiance than with exact directions), so it is kept:
                    // round-trip through the world-space point.
                    ps.d = (ps.o + ps.d) - ps.o;
#endif
#if RTX_DET
                    float u = fdiv(fi + rng_nextf(&ps.rng), fnx);  // Scene.cpp:145-146
                    float v = fdiv(fj + rng_nextf(&ps.rng), fny);
                    // Camera::getRay, Camera.h:24-26
                    ps.o = v3(sc.cam_origin);
                    ps.d = ((v3(sc.cam_ll) + v3(sc.cam_u) * u) + v3(sc.cam_v) * v) - ps.o;
#endif
                    ps.T = v3(1.0f, 1.0f, 1.0f);
                    ps.depth = 0;
                    segs = 0;
                    active = true;
                }
            }
        }
        // (no lane with an item although the pool is not spent: only slots outside the image were
        // drawn on a ragged edge tile; the body below is harmless without items and the loop
        // keeps a single back edge)
        // Linear scan: no branch around the body.  A lane without an item (only while the pool
        // drains, or on a ragged edge tile) follows whatever ray it holds and its results are
        // thrown away; what that costs is nothing the wave would not execute anyway, and the
        // path state is then carried round the loop without a copy in and out of a conditional
        // region.  BVH: the traversal is per lane, an idle lane stays out of it.
        if (kBvh ? active : true) {
            float t_hit;
            uint32_t prim_hit;
#if RTX_DET
            const bool hit = scene_query<kBvh>(sc, ps.o, ps.d, kTMin, kFltMax, t_hit, prim_hit);
#else
            // A wave whose rays all pass the scene by - the pixels of a tile that look past it - answers
            // its queries from the scene's bounds and skips the scan (the linear scan is wave-uniform:
            // it only pays when no lane needs it; the tree rejects such a ray at its root anyway).
            // Only camera rays come from outside the scene: the test runs when every ray of the wave is one.
            bool hit = false;
            // (the lanes that held a path before this turn's draw are those outside `want`, the paths
            // drawn just now are at depth 0: one ballot of one comparison)
            bool scan = kBvh || (rtx_ballot(ps.depth != 0u) & ~want) != 0ull;
            if (!scan) scan = rtx_ballot(active && ray_meets_scene(sc, ps.o, ps.d, kTMin)) != 0ull;
            if (scan) hit = scene_query<kBvh>(sc, ps.o, ps.d, kTMin, kFltMax, t_hit, prim_hit);
#endif
            ++segs;
            V3 L;
            bool ended;
            if (hit) {  // the winner's record lives inside this branch only
                HitRec rec;
                scene_finish<kBvh>(sc, prim_hit, t_hit, ps.o, ps.d, false, rec);
                ended = shade<kLightTree, kListLit>(sc, a.max_depth, true, rec, ps, L);
            } else {
                HitRec none;
                none.bits = 0u;
                ended = shade<kLightTree, kListLit>(sc, a.max_depth, false, none, ps, L);
            }
            if (ended && active) {
                const uint32_t slot = __builtin_amdgcn_ubfe(item, 0u, pixl);
                atomicAdd(&s_wave[wave].acc[0][slot], L.x);  // Scene.cpp:148
                atomicAdd(&s_wave[wave].acc[1][slot], L.y);
                atomicAdd(&s_wave[wave].acc[2][slot], L.z);
                rays += segs;
                if (kSamples) {
                    const uint32_t i = x0 + (slot & (tile_w - 1u)), j = y0 + (slot >> twl);
                    size_t k = (size_t)(j * a.width + i) * a.spp_count + (item >> pixl);
                    a.sample_rgb[3 * k + 0] = L.x;
                    a.sample_rgb[3 * k + 1] = L.y;
                    a.sample_rgb[3 * k + 2] = L.z;
                    if (a.sample_segments) a.sample_segments[k] = segs;
                }
                active = false;
            }
        }
    }

    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t sx = lane & (tile_w - 1u), sy = lane >> twl;
    if (a.accum && sx < tw && sy < th) {
        const uint32_t i = x0 + sx, j = y0 + sy;
        float* out = a.accum + ((size_t)(a.height - 1u - j) * a.width + i) * 3u;  // Scene.cpp:151
        const float r = s_wave[wave].acc[0][lane], g = s_wave[wave].acc[1][lane], b = s_wave[wave].acc[2][lane];
        if (a.accumulate) {
            out[0] += r;
            out[1] += g;
            out[2] += b;
        } else {
            out[0] = r;
            out[1] = g;
            out[2] = b;
        }
    }
    if (a.rays) {
        // wave-level sum, one atomic per wave
        uint32_t r = rays;
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
        if (lane == 0u) atomicAdd(a.rays, (unsigned long long)r);
    }
}

#if !RTX_DET
// BVH scenes, fast mode: the same pool of (pixel, sample) items per wave, but traversal and
// shading are decoupled.  A lane is idle, traversing, or holds a finished nearest-hit query;
// each iteration of the wave performs ONE traversal step (a node, or a leaf) for the lanes
// that are traversing (RTX_WALK_STEPS of them between two looks at the wave's state), and the
// shading / camera-ray stage only runs once kShadeBatch lanes wait for it (or nothing is left
// to traverse).  Traversal lengths of incoherent rays differ
// by an order of magnitude: in trace_kernel every lane waits for the longest of the wave's
// 64 queries before any of them is shaded, here a lane that is done is refilled while the
// others walk on, and the scattering code runs on a well-filled wave.
// Scheduling constants, measured on the 487-sphere field and on slabs of 2 000 - 100 000 items at 1080p:
// walk steps between two looks at the wave's state 4 / 8 / 16: 22.5 / 23.5 / 23.2 Grays/s (sphere field);
// lanes that must wait before a shading pass runs 16 / 24 / 32: 20.8 / 22.5 / 23.1 (sphere field), on trees of
// 8 192 nodes and more (kLongWalks; the longer the walks, the sooner the finished lanes should be shaded)
// 24 / 32: 14.8 / 14.6 (20 000 items), 8.53 / 8.39 (100 000 items; 16: 8.43, 40: 8.03), but 30.9 / 31.2 on
// 2 000 items; lanes that must hold a leaf record before the leaf code runs: 4 .. 32 measured, 8 best.
#ifndef RTX_SHADE_BATCH2
#define RTX_SHADE_BATCH2 32
#endif
#ifndef RTX_SHADE_BATCH2_LONG
#define RTX_SHADE_BATCH2_LONG 24
#endif
#ifndef RTX_WALK_STEPS2
#define RTX_WALK_STEPS2 8
#endif
#ifndef RTX_LEAF_BATCH
#define RTX_LEAF_BATCH 8
#endif
// One wave per block: a block's LDS and wave slots are free as soon as ITS pool is spent (the
// waves of a 256-thread block wait for the slowest of the four).  Six waves per SIMD are 24
// blocks per CU, which leaves 160 KB / 24 = 6 826 bytes of LDS for each: the per-lane stack is
// 5.5 KB and the pixel sums 0.75 KB (with 6 KB of stack, 6 912 bytes in all, only 22 - 23 blocks
// fitted: 5 - 7 % slower on the BVH scenes).
#ifndef RTX_BVH_BLOCK
#define RTX_BVH_BLOCK 64
#endif
#ifndef RTX_BVH_WAVES
#define RTX_BVH_WAVES 6  // minimum resident waves per SIMD the kernel is compiled for (80 VGPRs, none spilled; 5: 82)
#endif

// Development aid (-DRTX_BVH_STATS, tools/bvh_stats.py): per-wave event counts, added up in the
// eight 64-bit words behind the ray counter.
#ifdef RTX_BVH_STATS
#define RTX_STAT(i, n) (stat[i] += (n))
#else
#define RTX_STAT(i, n) ((void)0)
#endif

// The kernel walks the two-child tree of the deterministic mode (sc.bvh) and reads the fast-mode
// leaf records (sc.bvh_fast).  kLongWalks: see the scheduling constants above.
template <bool kLightTree, bool kSamples, bool kListLit = false, bool kLongWalks = false>
__global__ void __launch_bounds__(RTX_BVH_BLOCK, RTX_BVH_WAVES) trace_bvh_kernel(const TraceArgs a) {
#ifdef RTX_BVH_STATS
    uint32_t stat[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // wave-uniform
#endif
    __shared__ float s_acc[RTX_BVH_BLOCK / 64][3][64];
    // per lane: the deferred children, one per level
    __shared__ uint32_t s_stack2[kBvhStack][RTX_BVH_BLOCK];
    static_assert((sizeof(s_stack2) + sizeof(s_acc)) * 4 * RTX_BVH_WAVES <= 160 * 1024, "LDS must not bound the occupancy");
    const SceneView& sc = a.scene;
    const uint32_t tid = threadIdx.x, lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t twl = a.tile_w_log2, pixl = a.tile_w_log2 + a.tile_h_log2;
    const uint32_t tile_w = 1u << twl, tile_h = 1u << a.tile_h_log2, pix_mask = (1u << pixl) - 1u;
    constexpr uint32_t kSide = RTX_BVH_BLOCK == 256 ? 2u : 1u;
    const uint32_t tiles_x = (a.width + kSide * tile_w - 1u) / (kSide * tile_w);
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t x0 = (kSide * bx + (wave & (kSide - 1u))) * tile_w;
    const uint32_t y0 = (kSide * by + (wave >> (kSide - 1u))) * tile_h;
    if (x0 >= a.width || y0 >= a.height) return;  // wave-uniform
    const uint32_t tw = min(tile_w, a.width - x0), th = min(tile_h, a.height - y0);
    s_acc[wave][0][lane] = 0.0f;
    s_acc[wave][1][lane] = 0.0f;
    s_acc[wave][2][lane] = 0.0f;

    const uint32_t total = a.spp_count << pixl;
    uint32_t next = 0;
    uint32_t rays = 0, segs = 0, item = 0;
    enum : uint32_t { kIdle = 0u, kWalk = 1u, kDoneWalk = 2u, kNewRay = 3u };
    uint32_t phase = kIdle;
    PathState ps;
    ps.o = ps.d = ps.T = v3(0.0f, 0.0f, 0.0f);
    ps.depth = 0;
    ps.rng.state = 0;
    ps.rng.inc = 1;
    // The query in flight: the node or leaf the lane stands at, its stack pointer, the winner so far,
    // the ray's reciprocal direction and the ray on the tree's grid.
    uint32_t sp = 0, best = kNone;
    float closest = kFltMax;
    V3 inv = v3(0.0f, 0.0f, 0.0f), noi = inv;
    constexpr uint32_t kDone = 0xffffffffu;
    uint32_t ref = kDone;
    BvhRay br;
    br.A = br.B = inv;
    const float fnx = (float)a.width, fny = (float)a.height;

    for (;;) {
        const unsigned long long walking = rtx_ballot(phase == kWalk);
        const uint32_t n_done = (uint32_t)__popcll(rtx_ballot(phase == kDoneWalk));
        const unsigned long long idle = rtx_ballot(phase == kIdle);
        const bool pool_left = next < total;
        const uint32_t waiting = n_done + (pool_left ? (uint32_t)__popcll(idle) : 0u);
        constexpr uint32_t kShadeBatch = kLongWalks ? RTX_SHADE_BATCH2_LONG : RTX_SHADE_BATCH2;
        if (waiting != 0u && (walking == 0ull || waiting >= kShadeBatch)) {
            RTX_STAT(5, 1u);      // [5] shade / refill passes of the wave
            RTX_STAT(6, n_done);  // [6] ... lanes shaded
            if (phase == kDoneWalk) {  // Scene::color for the finished query
                ++segs;
                V3 L;
                bool ended;
                if (best != kNone) {  // the winner's record lives inside this branch only
                    HitRec rec;
                    finish_scan_hit<true>(sc, best, closest, ps.o, ps.d, false, rec);
                    ended = shade<kLightTree, kListLit>(sc, a.max_depth, true, rec, ps, L);
                } else {
                    HitRec none;
                    none.bits = 0u;
                    ended = shade<kLightTree, kListLit>(sc, a.max_depth, false, none, ps, L);
                }
                if (ended) {
                    const uint32_t slot = item & pix_mask;
                    atomicAdd(&s_acc[wave][0][slot], L.x);  // Scene.cpp:148
                    atomicAdd(&s_acc[wave][1][slot], L.y);
                    atomicAdd(&s_acc[wave][2][slot], L.z);
                    rays += segs;
                    if (kSamples) {  // per-sample outputs, as trace_kernel writes them
                        const uint32_t i = x0 + (slot & (tile_w - 1u)), j = y0 + (slot >> twl);
                        size_t k = (size_t)(j * a.width + i) * a.spp_count + (item >> pixl);
                        a.sample_rgb[3 * k + 0] = L.x;
                        a.sample_rgb[3 * k + 1] = L.y;
                        a.sample_rgb[3 * k + 2] = L.z;
                        if (a.sample_segments) a.sample_segments[k] = segs;
                    }
                    phase = kIdle;
                } else {
                    phase = kNewRay;
                }
            }
            const unsigned long long want = rtx_ballot(phase == kIdle);
            if (want != 0ull && next < total) {  // idle lanes draw the next items of the pool
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                const uint32_t mine = next + rank;
                next += (uint32_t)__popcll(want);
                if (phase == kIdle && mine < total) {
                    const uint32_t slot = mine & pix_mask;
                    const uint32_t sx = slot & (tile_w - 1u), sy = slot >> twl;
                    if (sx < tw && sy < th) {
                        item = mine;
                        const uint32_t i = x0 + sx, j = y0 + sy;
                        rng_init(&ps.rng, j * a.width + i, a.spp_begin + (mine >> pixl), a.seed);
                        float u = fdiv((float)i + rng_nextf(&ps.rng), fnx);  // Scene.cpp:145-146
                        float v = fdiv((float)j + rng_nextf(&ps.rng), fny);
                        ps.o = v3(sc.cam_origin);  // Camera::getRay, Camera.h:24-26
                        ps.d = ((v3(sc.cam_ll) + v3(sc.cam_u) * u) + v3(sc.cam_v) * v) - ps.o;
                        ps.T = v3(1.0f, 1.0f, 1.0f);
                        ps.depth = 0;
                        segs = 0;
                        phase = kNewRay;
                    }
                }
            }
            if (phase == kNewRay) {  // start the nearest-hit query of the new segment
                constexpr float kBig = 1.0e18f;  // see hit_bvh
                inv = v3(__builtin_amdgcn_fmed3f(frcp(ps.d.x), -kBig, kBig), __builtin_amdgcn_fmed3f(frcp(ps.d.y), -kBig, kBig),
                         __builtin_amdgcn_fmed3f(frcp(ps.d.z), -kBig, kBig));
                noi = v3(-ps.o.x * inv.x, -ps.o.y * inv.y, -ps.o.z * inv.z);
                br = bvh_ray_signed(sc, inv, noi);
                ref = 0u;  // root
                sp = 0u;
                closest = kFltMax;
                best = kNone;
                // the items outside the tree: every ray tests them (wave-uniform records)
                for (uint32_t k = 0; k < sc.n_bvh_big; ++k)
                    fast_leaf_item(sc, load_uniform(sc.bvh_fast, sc.bvh_big_first + k), ps.o, ps.d, inv, noi, kTMin,
                                   closest, best);
                phase = kWalk;
            }
        }
        if (rtx_ballot(phase != kIdle) == 0ull) {
            if (next >= total) break;  // pool spent, nothing in flight
            continue;                  // this draw fell outside the image (ragged tile): draw again
        }
#pragma unroll 1
        for (int step = 0; step < RTX_WALK_STEPS2; ++step) {
            RTX_STAT(4, 1u);  // [4] walk steps of the wave
            // A lane at a leaf tests its record; a lane at a node goes one node down.  Lanes at a leaf
            // hold it until kLeafBatch of them have one (or nobody is left at a node), so that the
            // leaf code - four record types, each its own branch - runs on a fuller wave.
            // (each ballot is of ONE comparison - its condition mask is the ballot - and the masks are
            // combined on the scalar unit: the ballot of a compound condition costs two more vector
            // instructions, which turn the combined mask into 0 / 1 per lane and compare that again)
            const bool holds_items = (ref & kBvhLeaf) != 0u;
            const bool at_leaf = phase == kWalk && holds_items;
            const bool at_node = phase == kWalk && !at_leaf;
            const unsigned long long walk_mask = rtx_ballot(phase == kWalk), item_mask = rtx_ballot(holds_items);
            const unsigned long long nodes = walk_mask & ~item_mask;
            const uint32_t n_leaves = (uint32_t)__popcll(walk_mask & item_mask);
            if (nodes == 0ull && n_leaves == 0u) break;
            RTX_STAT(7, (uint32_t)__popcll(rtx_ballot(phase == kDoneWalk)));  // [7] lanes waiting to be shaded
            RTX_STAT(8, (uint32_t)__popcll(rtx_ballot(phase == kIdle)));      // [8] lanes without an item
            RTX_STAT(9, n_leaves);                                           // [9] lanes holding leaf items
            if (nodes != 0ull) {
                RTX_STAT(0, 1u);                            // [0] node steps of the wave
                RTX_STAT(1, (uint32_t)__popcll(nodes));     // [1] ... of its lanes
            }
            constexpr uint32_t kLeafBatch = RTX_LEAF_BATCH;
            const bool leaves_now = n_leaves >= kLeafBatch || (nodes == 0ull && n_leaves != 0u);
            if (leaves_now) {
                RTX_STAT(2, 1u);        // [2] item steps of the wave
                RTX_STAT(3, n_leaves);  // [3] ... of its lanes
            }
            if (at_node) {
                // Without branches: the top of the stack is fetched beside the node whether or not this
                // step pops it, and the farther child is written above the top whether or not both
                // children are met (sp only moves past it if they are; rtx_upload_scene refuses a tree
                // deeper than the stack, so that row exists).
                const uint32_t below = s_stack2[sp == 0u ? 0u : sp - 1u][tid];
                const NodeWords N = load_node(sc, ref);
                float n0, n1;
                const bool h0 = slab_signed(N.a.x, N.a.y, N.a.z, br, kTMin, closest, n0);
                const bool h1 = slab_signed(N.a.w, N.b.x, N.b.y, br, kTMin, closest, n1);
                const bool both = h0 && h1, any = h0 || h1;
                const bool take0 = h0 && (!h1 || n0 <= n1);  // the nearer child first
                s_stack2[sp][tid] = take0 ? N.b.w : N.b.z;
                const bool pop = !any && sp != 0u;
                ref = any ? (take0 ? N.b.z : N.b.w) : (sp != 0u ? below : kDone);
                sp = sp + (both ? 1u : 0u) - (pop ? 1u : 0u);
                if (ref == kDone) phase = kDoneWalk;
            }
            if (leaves_now && at_leaf) {
                // (a leaf of this tree is one record: rtx_bvh.cpp builds it with kLeafItems = 1)
                const uint32_t below = s_stack2[sp == 0u ? 0u : sp - 1u][tid];  // fetched beside the record
                fast_leaf_item(sc, load_fast_item(sc.bvh_fast, ref & kBvhLeafFirst), ps.o, ps.d, inv, noi, kTMin, closest, best);
                ref = sp != 0u ? below : kDone;
                sp -= sp != 0u ? 1u : 0u;
                if (ref == kDone) phase = kDoneWalk;
            }
        }
    }

    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t sx = lane & (tile_w - 1u), sy = lane >> twl;
    if (a.accum && sx < tw && sy < th) {
        const uint32_t i = x0 + sx, j = y0 + sy;
        float* out = a.accum + ((size_t)(a.height - 1u - j) * a.width + i) * 3u;  // Scene.cpp:151
        const float r = s_acc[wave][0][lane], g = s_acc[wave][1][lane], b = s_acc[wave][2][lane];
        if (a.accumulate) {
            out[0] += r;
            out[1] += g;
            out[2] += b;
        } else {
            out[0] = r;
            out[1] = g;
            out[2] = b;
        }
    }
    if (a.rays) {
        uint32_t r = rays;
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
        if (lane == 0u) atomicAdd(a.rays, (unsigned long long)r);
#ifdef RTX_BVH_STATS
        if (lane == 0u)
            for (int k = 0; k < 12; ++k) atomicAdd(a.rays + 1 + k, (unsigned long long)stat[k]);
#endif
    }
}
#undef RTX_STAT
#endif

// Shape::hit of the world for arbitrary rays.
template <bool kBvh>
__global__ void __launch_bounds__(256) hit_kernel(const HitArgs a) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    // every lane runs the (wave-uniform) scan; out-of-range lanes repeat ray 0
    uint32_t kk = k < a.n ? k : 0u;
    V3 o = v3(a.origins + 3u * kk), d = v3(a.directions + 3u * kk);
    HitRec rec;
    rec.t = 0.0f;
    rec.p = rec.n = v3(0.0f, 0.0f, 0.0f);
    rec.prim = kNone;
    rec.u = rec.v = 0.0f;
    bool hit = scene_hit<kBvh>(a.scene, o, d, a.t0, a.t1, rec, true);
    if (k < a.n) {
        RtxHit* out = reinterpret_cast<RtxHit*>(a.hits) + k;
        out->hit = hit ? 1 : 0;
        const uint32_t prim = hit ? original_prim(a.scene, rec.prim, kBvh) : 0u;
        out->prim = hit ? (int32_t)prim : -1;
        out->t = hit ? rec.t : 0.0f;
        out->p[0] = hit ? rec.p.x : 0.0f;
        out->p[1] = hit ? rec.p.y : 0.0f;
        out->p[2] = hit ? rec.p.z : 0.0f;
        out->n[0] = hit ? rec.n.x : 0.0f;
        out->n[1] = hit ? rec.n.y : 0.0f;
        out->n[2] = hit ? rec.n.z : 0.0f;
        out->material = hit ? (int32_t)a.scene.prims[prim].material : -1;
        out->uv[0] = hit ? rec.u : 0.0f;
        out->uv[1] = hit ? rec.v : 0.0f;
    }
}

}  // namespace RTX_NS
