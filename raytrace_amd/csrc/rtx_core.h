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
//       same algorithm and random stream, arithmetic free to fuse and approximate.
//
// Reference citations are relative to Raytrace_C++/Src.
#pragma once

#include <hip/hip_runtime.h>
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

namespace RTX_NS {

using namespace rtxd;

constexpr float kPi = 3.14159265359f;   // Util.h:2
constexpr float kPi2 = 6.28318530718f;  // Util.h:3
constexpr float kTMin = 0.001f;         // Scene.cpp:105, Sphere.cpp:40, Rect.cpp:39
constexpr float kFltMax = 3.402823466e+38f;
constexpr uint32_t kNone = 0xffffffffu;

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
RTX_DEV float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

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

// Wave-uniform record fetch: the index is the same in every lane, so tell the
// compiler (scalar loads, SGPR operands).
template <typename T>
RTX_DEV T load_uniform(const T* base, uint32_t i) {
    return base[__builtin_amdgcn_readfirstlane(i)];
}

struct HitRec {
    float t;
    V3 p, n;
    uint32_t prim;
    uint32_t material;
};

// ---------------------------------------------------------------------------
// primitive tests.  Both return the accepted t or leave `t_out` untouched.
// ---------------------------------------------------------------------------

// Rect::hit, Rect.cpp:16-25.  A NaN t or NaN x / y passes both rejection tests,
// exactly as in the reference (the comparisons are written as rejections).
RTX_DEV bool rect_test(const Prim& P, V3 o, V3 d, V3 inv_d, float t0, float t1, float& t_out) {
    int xi, yi, zi;
    if (P.kind == RTX_PRIM_RECT_XY) { xi = 0; yi = 1; zi = 2; }
    else if (P.kind == RTX_PRIM_RECT_XZ) { xi = 0; yi = 2; zi = 1; }
    else { xi = 1; yi = 2; zi = 0; }
#if RTX_DET
    float t = (P.e - comp(o, zi)) / comp(d, zi);
    (void)inv_d;
    if (t < t0 || t > t1) return false;
    float x = comp(o, xi) + t * comp(d, xi);
    float y = comp(o, yi) + t * comp(d, yi);
    if (x < P.a || x > P.b || y < P.c || y > P.d) return false;
#else
    float t = (P.e - comp(o, zi)) * comp(inv_d, zi);
    float x = comp(o, xi) + t * comp(d, xi);
    float y = comp(o, yi) + t * comp(d, yi);
    bool ok = (t >= t0) & (t <= t1) & (x >= P.a) & (x <= P.b) & (y >= P.c) & (y <= P.d);
    if (!ok) return false;
#endif
    t_out = t;
    return true;
}

// Sphere::hit, Sphere.cpp:8-33 (roots only; p / n are derived by the caller).
RTX_DEV bool sphere_test(float cx, float cy, float cz, float radius, V3 o, V3 d, float t0, float t1,
                         float& t_out) {
    V3 oc = o - v3(cx, cy, cz);
    float a = dot(d, d);
    float b = 2.0f * dot(oc, d);
    float c = dot(oc, oc) - radius * radius;
    float D = b * b - (4.0f * a) * c;
    if (D > 0.0f) {
        float root = fsqrt(D);
#if RTX_DET
        float temp = (-b - root) / (2.0f * a);
        if (temp < t1 && temp > t0) { t_out = temp; return true; }
        temp = (-b + root) / (2.0f * a);
        if (temp < t1 && temp > t0) { t_out = temp; return true; }
#else
        float inv2a = frcp(2.0f * a);
        float ta = (-b - root) * inv2a;
        float tb = (-b + root) * inv2a;
        bool oka = (ta < t1) & (ta > t0);
        bool okb = (tb < t1) & (tb > t0);
        if (oka | okb) { t_out = oka ? ta : tb; return true; }
#endif
    }
    return false;
}

// ---------------------------------------------------------------------------
// nearest hit: linear scan in list order (ShapeList::hit, ShapeList.cpp:5-17)
// ---------------------------------------------------------------------------

// Finish the record of the winning primitive of one group in the group's local
// frame, then carry it out through the group's wrappers (innermost first):
// Rotate::hit (Rotate.cpp:12-13) rotates p and n, Translate::hit
// (Translate.cpp:9) offsets p, FlipNormals::hit (FlipNormals.cpp:8) negates n.
RTX_DEV void finish_hit(const SceneView& sc, const Group& G, uint32_t prim, float t, V3 ro, V3 rd,
                        HitRec& rec) {
    const Prim W = sc.prims[prim];  // per-lane
    V3 p = ro + rd * t;             // Ray::at, Ray.h:12
    V3 n;
    if (W.kind == RTX_PRIM_SPHERE) {
        n = vdivs(p - v3(W.a, W.b, W.c), W.d);  // Sphere.cpp:19
    } else if (W.kind == RTX_PRIM_RECT_XY) {
        n = v3(0.0f, 0.0f, 1.0f);
    } else if (W.kind == RTX_PRIM_RECT_XZ) {
        n = v3(0.0f, 1.0f, 0.0f);
    } else {
        n = v3(1.0f, 0.0f, 0.0f);
    }
    if (W.flip) n = -n;
    for (uint32_t s = G.step_end; s > G.step_begin; --s) {
        const Step S = load_uniform(sc.steps, s - 1);
        if (S.kind == RTX_STEP_ROTATE) {
            p = qrotate(S.x, S.y, S.z, S.w, p);
            n = qrotate(S.x, S.y, S.z, S.w, n);
        } else {
            p = p + v3(S.x, S.y, S.z);
        }
    }
    rec.t = t;
    rec.p = p;
    rec.n = n;
    rec.prim = prim;
    rec.material = W.material;
}

// Scan the primitives of one group with the ray already in the group's frame.
RTX_DEV void scan_group(const SceneView& sc, const Group& G, V3 ro, V3 rd, float t0, float& closest,
                        uint32_t& best) {
    V3 inv_d = v3(0.0f, 0.0f, 0.0f);
#if !RTX_DET
    inv_d = v3(frcp(rd.x), frcp(rd.y), frcp(rd.z));
#endif
    for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) {
        const Prim P = load_uniform(sc.prims, p);
        float t;
        bool ok;
        if (P.kind == RTX_PRIM_SPHERE) ok = sphere_test(P.a, P.b, P.c, P.d, ro, rd, t0, closest, t);
        else ok = rect_test(P, ro, rd, inv_d, t0, closest, t);
        if (ok) {
            closest = t;
            best = p;
        }
    }
}

// Bring a ray into a group's frame: Translate::hit (Translate.cpp:7) subtracts
// the offset from the origin, Rotate::hit (Rotate.cpp:7-9) rotates origin and
// direction by the conjugate quaternion; outermost wrapper first.
RTX_DEV void enter_group(const SceneView& sc, const Group& G, V3& ro, V3& rd) {
    for (uint32_t s = G.step_begin; s < G.step_end; ++s) {
        const Step S = load_uniform(sc.steps, s);
        if (S.kind == RTX_STEP_ROTATE) {
            ro = qrotate(-S.x, -S.y, -S.z, S.w, ro);
            rd = qrotate(-S.x, -S.y, -S.z, S.w, rd);
        } else {
            ro = ro - v3(S.x, S.y, S.z);
        }
    }
}

RTX_DEV bool hit_linear(const SceneView& sc, V3 o, V3 d, float t0, float t1, HitRec& rec) {
    float closest = t1;
    bool any = false;
    for (uint32_t g = 0; g < sc.n_groups; ++g) {
        const Group G = load_uniform(sc.groups, g);
        V3 ro = o, rd = d;
        enter_group(sc, G, ro, rd);
        uint32_t best = kNone;
        scan_group(sc, G, ro, rd, t0, closest, best);
        if (best != kNone) {
            finish_hit(sc, G, best, closest, ro, rd, rec);
            any = true;
        }
    }
    return any;
}

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
constexpr int kBvhStack = 32;

RTX_DEV bool tie_wins(uint32_t cand, bool cand_is_rect, uint32_t cur, bool cur_is_rect) {
    if (cand_is_rect && cur_is_rect) return cand > cur;
    if (cand_is_rect != cur_is_rect) return cand_is_rect;
    return cand < cur;
}

RTX_DEV bool hit_bvh(const SceneView& sc, V3 o, V3 d, float t0, float t1, HitRec& rec) {
    uint32_t stack[kBvhStack];
    int sp = 0;
    float closest = t1;
    uint32_t best = kNone, best_group = 0;
    bool best_is_rect = false;
    V3 best_ro = o, best_rd = d;
    V3 inv = v3(frcp(d.x), frcp(d.y), frcp(d.z));
    uint32_t node = 0;
    for (;;) {
        const BvhNode N = sc.bvh[node];
        // slab test, conservative: boxes are padded by the builder
        float tx0 = (N.lo[0] - o.x) * inv.x, tx1 = (N.hi[0] - o.x) * inv.x;
        float ty0 = (N.lo[1] - o.y) * inv.y, ty1 = (N.hi[1] - o.y) * inv.y;
        float tz0 = (N.lo[2] - o.z) * inv.z, tz1 = (N.hi[2] - o.z) * inv.z;
        float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), t0));
        float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), closest));
        bool overlap = tn <= tf;
        if (overlap) {
            if (N.count == 0) {
                stack[sp++] = N.left + 1;
                node = N.left;
                continue;
            }
            for (uint32_t k = 0; k < N.count; ++k) {
                uint32_t g = sc.bvh_items[N.left + k];
                const Group G = sc.groups[g];
                V3 ro = o, rd = d;
                for (uint32_t s = G.step_begin; s < G.step_end; ++s) {
                    const Step S = sc.steps[s];
                    if (S.kind == RTX_STEP_ROTATE) {
                        ro = qrotate(-S.x, -S.y, -S.z, S.w, ro);
                        rd = qrotate(-S.x, -S.y, -S.z, S.w, rd);
                    } else {
                        ro = ro - v3(S.x, S.y, S.z);
                    }
                }
                V3 inv_d = v3(0.0f, 0.0f, 0.0f);
#if !RTX_DET
                inv_d = v3(frcp(rd.x), frcp(rd.y), frcp(rd.z));
#endif
                for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) {
                    const Prim P = sc.prims[p];
                    float t;
                    bool is_rect = P.kind != RTX_PRIM_SPHERE;
                    bool ok;
                    // candidate t of this primitive alone, window (t0, closest] inclusive
                    // so that ties reach the explicit rule below
                    if (is_rect) {
                        ok = rect_test(P, ro, rd, inv_d, t0, closest, t);
                    } else {
                        ok = sphere_test(P.a, P.b, P.c, P.d, ro, rd, t0, t1, t) && t <= closest;
                    }
                    if (ok) {
                        bool take = t < closest || best == kNone ||
                                    tie_wins(p, is_rect, best, best_is_rect);
                        if (take) {
                            closest = t;
                            best = p;
                            best_is_rect = is_rect;
                            best_group = g;
                            best_ro = ro;
                            best_rd = rd;
                        }
                    }
                }
            }
        }
        if (sp == 0) break;
        node = stack[--sp];
    }
    if (best == kNone) return false;
    const Group G = sc.groups[best_group];
    // finish_hit() expects wave-uniform steps; do the per-lane equivalent here
    const Prim W = sc.prims[best];
    V3 p = best_ro + best_rd * closest;
    V3 n;
    if (W.kind == RTX_PRIM_SPHERE) n = vdivs(p - v3(W.a, W.b, W.c), W.d);
    else if (W.kind == RTX_PRIM_RECT_XY) n = v3(0.0f, 0.0f, 1.0f);
    else if (W.kind == RTX_PRIM_RECT_XZ) n = v3(0.0f, 1.0f, 0.0f);
    else n = v3(1.0f, 0.0f, 0.0f);
    if (W.flip) n = -n;
    for (uint32_t s = G.step_end; s > G.step_begin; --s) {
        const Step S = sc.steps[s - 1];
        if (S.kind == RTX_STEP_ROTATE) {
            p = qrotate(S.x, S.y, S.z, S.w, p);
            n = qrotate(S.x, S.y, S.z, S.w, n);
        } else {
            p = p + v3(S.x, S.y, S.z);
        }
    }
    rec.t = closest;
    rec.p = p;
    rec.n = n;
    rec.prim = best;
    rec.material = W.material;
    return true;
}

template <bool kBvh>
RTX_DEV bool scene_hit(const SceneView& sc, V3 o, V3 d, float t0, float t1, HitRec& rec) {
    if (kBvh) return hit_bvh(sc, o, d, t0, t1, rec);
    return hit_linear(sc, o, d, t0, t1, rec);
}

// ---------------------------------------------------------------------------
// textures (Texture.h:5): ColorTexture.h:11-13, CheckerTexture.cpp:3-11
// ---------------------------------------------------------------------------
RTX_DEV V3 texture_value(const SceneView& sc, uint32_t tex, V3 p) {
    for (int level = 0; level < 16; ++level) {
        const Texture T = sc.textures[tex];
        if (T.kind == RTX_TEX_COLOR) return v3(T.r, T.g, T.b);
        float sines = (fsin(T.r * p.x) * fsin(T.r * p.y)) * fsin(T.r * p.z);
        tex = sines < 0.0f ? T.odd : T.even;
    }
    return v3(0.0f, 0.0f, 0.0f);
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
    V3 a = (__builtin_fabsf(b.w.x) > 0.9f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
    b.v = normalize(cross(b.w, a));
    b.u = cross(b.w, b.v);
    return b;
}
RTX_DEV V3 onb_local(const Onb& b, V3 a) {  // ONB.h:24-26
    return (b.u * a.x + b.v * a.y) + b.w * a.z;
}

// ---------------------------------------------------------------------------
// light sampling: ShapeList / Rect / Sphere :: random and :: pdf_value
// ---------------------------------------------------------------------------

// Shape::random of one light entry (Rect.cpp:50-68, Sphere.cpp:52-58, Shape.h:9)
RTX_DEV V3 light_random_one(const Light& E, V3 o, RtxRng& rng) {
    if (E.kind == RTX_LIGHT_RECT_XZ) {
        float x = E.a + rtx_rng_nextf(&rng) * (E.b - E.a);
        float y = E.c + rtx_rng_nextf(&rng) * (E.d - E.c);
        return v3(x, E.e, y) - o;
    }
    if (E.kind == RTX_LIGHT_SPHERE) {
        V3 direction = v3(E.a, E.b, E.c) - o;
        float dd = length_sqr(direction);
        Onb uvw = onb_from_w(direction);
        // random_to_sphere, Util.h:96-107
        float r1 = rtx_rng_nextf(&rng);
        float r2 = rtx_rng_nextf(&rng);
        float r2r = E.d * E.d;
        float rr = dd < r2r ? dd : r2r;  // std::min(pow2(radius), distance_squared)
        float cos_theta_max = fsqrt(1.0f - rr * frcp(dd));
        float z = 1.0f - r2 * (1.0f - cos_theta_max);
        float sqrtz = fsqrt(1.0f - z * z);
        float sn, cs;
        sincos_2pi(r1, sn, cs);
        return onb_local(uvw, v3(cs * sqrtz, sn * sqrtz, z));
    }
    return v3(1.0f, 0.0f, 0.0f);
}

// ShapePdf::generate -> Shape::random of the light shape (ShapePdf.h:15-17);
// for a list one draw picks the entry (ShapeList.cpp:28-35).
RTX_DEV V3 light_random(const SceneView& sc, V3 o, RtxRng& rng) {
    uint32_t index = 0;
    if (sc.light_is_list) {
        float r = rtx_rng_nextf(&rng);
        index = (uint32_t)(r * (float)sc.n_lights);
        if (index >= sc.n_lights) index = sc.n_lights - 1;
    }
    const Light E = sc.lights[index];  // per-lane
    return light_random_one(E, o, rng);
}

// Shape::pdf_value of one light entry (Rect.cpp:36-48, Sphere.cpp:38-50, Shape.h:8)
RTX_DEV float light_pdf_one(const Light& E, V3 o, V3 v) {
    if (E.kind == RTX_LIGHT_RECT_XZ) {
        Prim P;
        P.a = E.a; P.b = E.b; P.c = E.c; P.d = E.d; P.e = E.e;
        P.kind = RTX_PRIM_RECT_XZ;
        float t;
        V3 inv = v3(frcp(v.x), frcp(v.y), frcp(v.z));
        if (!rect_test(P, o, v, inv, kTMin, kFltMax, t)) return 0.0f;
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
        float cos_theta_max = fsqrt(1.0f - rr * frcp(dd));
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

RTX_DEV bool shade(const SceneView& sc, uint32_t max_depth, bool hit, const HitRec& rec, PathState& ps,
                   V3& L) {
    if (!hit) {  // Scene.cpp:131
        L = vmul(ps.T, v3(sc.background));
        return true;
    }
    const Material M = sc.materials[rec.material];
    if (M.kind == RTX_MAT_DIFFUSE_LIGHT) {  // DiffuseLight.h:17-28: emits on the front face, never scatters
        V3 e = v3(0.0f, 0.0f, 0.0f);
        if (dot(rec.n, ps.d) < 0.0f) e = texture_value(sc, M.texture, rec.p);
        L = vmul(ps.T, e);
        return true;
    }
    // the other materials emit nothing (Material.h:10)
    L = v3(0.0f, 0.0f, 0.0f);
    if (ps.depth >= max_depth) return true;  // Scene.cpp:108

    if (M.kind == RTX_MAT_LAMBERTIAN) {
        V3 albedo = texture_value(sc, M.texture, rec.p);  // Lambertian.cpp:17
        // MixturePdf::generate, MixturePdf.h:15-22
        V3 dir;
        if (rtx_rng_nextf(&ps.rng) < 0.5f) {
            dir = light_random(sc, rec.p, ps.rng);
        } else {
            // CosinePdf::generate (CosinePdf.cpp:16-20), random_cosine_direction (Util.h:86-94)
            Onb uvw = onb_from_w(rec.n);
            float r1 = rtx_rng_nextf(&ps.rng);
            float r2 = rtx_rng_nextf(&ps.rng);
            float z = fsqrt(1.0f - r2);
            float sn, cs;
            sincos_2pi(r1, sn, cs);
            float sr2 = fsqrt(r2);
            dir = onb_local(uvw, v3(cs * sr2, sn * sr2, z));
        }
        // MixturePdf::value, MixturePdf.h:9-13
        float pdf0 = light_pdf(sc, rec.p, dir);
        float cosine = dot(normalize(dir), rec.n);  // CosinePdf.cpp:7
        float pdf1 = cosine > 0.0f ? fdiv(cosine, kPi) : 0.0f;
        float pdf_value = 0.5f * pdf0 + 0.5f * pdf1;
        if (!(pdf_value > 0.0f)) return true;  // Scene.cpp:117,122-124
        // Lambertian::scattering_pdf, Lambertian.cpp:23-25: std::max(cos, 0) / PI
        float c0 = cosine < 0.0f ? 0.0f : cosine;
        float spdf = fdiv(c0, kPi);
        ps.T = vdivs(vmul(albedo * spdf, ps.T), pdf_value);  // Scene.cpp:119-120
        ps.o = rec.p;
        ps.d = dir;
        ps.depth += 1;
        return false;
    }
    if (M.kind == RTX_MAT_METAL) {  // Metal.cpp:9-17
        V3 reflected = reflect(normalize(ps.d), rec.n);
        V3 q;
        do {  // random_in_unit_sphere, Util.h:32-38; runs even when fuzz == 0.
            // The oracle build (g++) evaluates the three draws of random_vector()
            // (Util.h:28-30) right to left: z, y, x.
            float rz = rtx_rng_nextf(&ps.rng);
            float ry = rtx_rng_nextf(&ps.rng);
            float rx = rtx_rng_nextf(&ps.rng);
            q = v3(rx, ry, rz) * 2.0f - v3(1.0f, 1.0f, 1.0f);
        } while (length_sqr(q) >= 1.0f);
        reflected = reflected + q * M.param;
        V3 albedo = texture_value(sc, M.texture, rec.p);
        if (!(dot(reflected, rec.n) > 0.0f)) return true;
        ps.T = vmul(albedo, ps.T);  // Scene.cpp:110
        ps.o = rec.p;
        ps.d = reflected;
        ps.depth += 1;
        return false;
    }
    // Dielectric, Dielectric.cpp:7-45
    {
        float ri = M.param;
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
        V3 dir = (rtx_rng_nextf(&ps.rng) < reflect_prob) ? reflected : refracted;
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

// Pixel owned by this lane: 8x8 tile per wave, 2x2 waves per 256-thread block.
RTX_DEV void lane_pixel(uint32_t tiles_x, uint32_t& i, uint32_t& j) {
    uint32_t lane = threadIdx.x & 63u;
    uint32_t wave = threadIdx.x >> 6;
    uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    i = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    j = by * 16u + (wave >> 1) * 8u + (lane >> 3);
}

// The loops of Scene::render (Scene.cpp:140-149) around the integrator.  Each lane
// owns one pixel and walks its samples in order; a lane whose path has ended starts
// its next sample at once, so lanes of a wave are at different samples / depths but
// all execute the same nearest-hit query each iteration.
template <bool kBvh, bool kSamples>
__global__ void __launch_bounds__(256) trace_kernel(const TraceArgs a) {
    const SceneView& sc = a.scene;
    uint32_t i, j;
    lane_pixel((a.width + 15u) / 16u, i, j);
    if (i >= a.width || j >= a.height) return;
    const uint32_t pixel = j * a.width + i;

    V3 acc = v3(0.0f, 0.0f, 0.0f);
    uint32_t rays = 0;
    uint32_t s = 0;  // samples started
    uint32_t segs = 0;
    bool active = false;
    PathState ps;
    ps.o = ps.d = ps.T = v3(0.0f, 0.0f, 0.0f);
    ps.depth = 0;
    ps.rng.state = 0;
    ps.rng.inc = 1;
    const float fi = (float)i, fj = (float)j;
    const float fnx = (float)a.width, fny = (float)a.height;

    for (;;) {
        if (!active) {
            if (s >= a.spp_count) break;
            rtx_rng_init(&ps.rng, pixel, a.spp_begin + s, a.seed);
            float u = fdiv(fi + rtx_rng_nextf(&ps.rng), fnx);  // Scene.cpp:145-146
            float v = fdiv(fj + rtx_rng_nextf(&ps.rng), fny);
            // Camera::getRay, Camera.h:24-26
            ps.o = v3(sc.cam_origin);
            ps.d = ((v3(sc.cam_ll) + v3(sc.cam_u) * u) + v3(sc.cam_v) * v) - ps.o;
            ps.T = v3(1.0f, 1.0f, 1.0f);
            ps.depth = 0;
            segs = 0;
            active = true;
            ++s;
        }
        HitRec rec;
        bool hit = scene_hit<kBvh>(sc, ps.o, ps.d, kTMin, kFltMax, rec);
        ++segs;
        V3 L;
        if (shade(sc, a.max_depth, hit, rec, ps, L)) {
            acc = acc + L;  // Scene.cpp:148
            rays += segs;
            if (kSamples) {
                size_t k = (size_t)pixel * a.spp_count + (s - 1);
                a.sample_rgb[3 * k + 0] = L.x;
                a.sample_rgb[3 * k + 1] = L.y;
                a.sample_rgb[3 * k + 2] = L.z;
                if (a.sample_segments) a.sample_segments[k] = segs;
            }
            active = false;
        }
    }

    if (a.accum) {
        float* out = a.accum + ((size_t)(a.height - 1u - j) * a.width + i) * 3u;  // Scene.cpp:151
        if (a.accumulate) {
            out[0] += acc.x;
            out[1] += acc.y;
            out[2] += acc.z;
        } else {
            out[0] = acc.x;
            out[1] = acc.y;
            out[2] = acc.z;
        }
    }
    if (a.rays) {
        // wave-level sum, one atomic per wave
        uint32_t r = rays;
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
        if ((threadIdx.x & 63u) == 0u) atomicAdd(a.rays, (unsigned long long)r);
    }
}

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
    rec.material = kNone;
    bool hit = scene_hit<kBvh>(a.scene, o, d, a.t0, a.t1, rec);
    if (k < a.n) {
        RtxHit* out = reinterpret_cast<RtxHit*>(a.hits) + k;
        out->hit = hit ? 1 : 0;
        out->prim = hit ? (int32_t)rec.prim : -1;
        out->t = hit ? rec.t : 0.0f;
        out->p[0] = hit ? rec.p.x : 0.0f;
        out->p[1] = hit ? rec.p.y : 0.0f;
        out->p[2] = hit ? rec.p.z : 0.0f;
        out->n[0] = hit ? rec.n.x : 0.0f;
        out->n[1] = hit ? rec.n.y : 0.0f;
        out->n[2] = hit ? rec.n.z : 0.0f;
        out->material = hit ? (int32_t)rec.material : -1;
    }
}

}  // namespace RTX_NS
