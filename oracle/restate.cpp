/*
 * restate.cpp — CPU restatement of the reference path tracer's algorithm over the
 * flattened scene description (include/rtx.h), in plain C++17.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the renderer links, loads or calls this
 * file; it exists so that tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py have an independent checker that can travel to machines where the
 * reference tree itself is absent.  It is pinned against the reference tracer
 * (oracle/_ref, built from the unmodified reference sources) by the golden
 * vectors under tests/golden/ and, where oracle/_ref is present, directly.
 *
 * It is written independently of the GPU code (recursive integrator, one
 * primitive at a time, no shared headers except the random-stream / sincos
 * contract of include/rtx_det.h) and follows the reference function by function;
 * each function cites the reference lines it restates (paths relative to
 * Raytrace_C++/Src, vectormath to external/sce_vectormath-master/include/
 * vectormath/scalar/cpp).
 *
 * The arithmetic type is a template parameter: `float` for rendering, `Counted`
 * for the op-counting build that yields the algorithmic flops per ray used by the
 * roofline figure (rule: every fp32 add / sub / mul / div / sqrt / sin / cos the
 * reference performs counts 1; compares, negation, abs, selects and the integer
 * random stream count 0).
 */
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "rtx.h"
#include "rtx_det.h"

namespace {

// ---------------------------------------------------------------------------
// arithmetic types
// ---------------------------------------------------------------------------
thread_local unsigned long long g_flops = 0;

struct Counted {
    float v;
    Counted() : v(0) {}
    Counted(float x) : v(x) {}
    explicit operator float() const { return v; }
};
inline Counted operator+(Counted a, Counted b) { ++g_flops; return Counted(a.v + b.v); }
inline Counted operator-(Counted a, Counted b) { ++g_flops; return Counted(a.v - b.v); }
inline Counted operator*(Counted a, Counted b) { ++g_flops; return Counted(a.v * b.v); }
inline Counted operator/(Counted a, Counted b) { ++g_flops; return Counted(a.v / b.v); }
inline Counted operator-(Counted a) { return Counted(-a.v); }
inline bool operator<(Counted a, Counted b) { return a.v < b.v; }
inline bool operator>(Counted a, Counted b) { return a.v > b.v; }
inline bool operator>=(Counted a, Counted b) { return a.v >= b.v; }

inline float raw(float x) { return x; }
inline float raw(Counted x) { return x.v; }
inline float f_sqrt(float x) { return ::sqrtf(x); }
inline Counted f_sqrt(Counted x) { ++g_flops; return Counted(::sqrtf(x.v)); }
inline float f_abs(float x) { return ::fabsf(x); }
inline Counted f_abs(Counted x) { return Counted(::fabsf(x.v)); }
inline void f_sincos(float x, float& s, float& c) { rtx_sincosf(x, &s, &c); }
inline void f_sincos(Counted x, Counted& s, Counted& c) { g_flops += 2; rtx_sincosf(x.v, &s.v, &c.v); }
inline float f_sin(float x) { return rtx_sinf(x); }
inline Counted f_sin(Counted x) { ++g_flops; return Counted(rtx_sinf(x.v)); }

const float kPI = 3.14159265359f;   // Util.h:2
const float kPI2 = 6.28318530718f;  // Util.h:3

template <class F>
struct Vec {
    F x, y, z;
    Vec() : x(0.0f), y(0.0f), z(0.0f) {}
    Vec(F a, F b, F c) : x(a), y(b), z(c) {}
    F at(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
// vec_aos.h:263-353
template <class F> Vec<F> operator+(const Vec<F>& a, const Vec<F>& b) { return Vec<F>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class F> Vec<F> operator-(const Vec<F>& a, const Vec<F>& b) { return Vec<F>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class F> Vec<F> operator-(const Vec<F>& a) { return Vec<F>(-a.x, -a.y, -a.z); }
template <class F> Vec<F> operator*(const Vec<F>& a, F s) { return Vec<F>(a.x * s, a.y * s, a.z * s); }
template <class F> Vec<F> operator/(const Vec<F>& a, F s) { return Vec<F>(a.x / s, a.y / s, a.z / s); }
template <class F> Vec<F> mul_per_elem(const Vec<F>& a, const Vec<F>& b) { return Vec<F>(a.x * b.x, a.y * b.y, a.z * b.z); }
// vec_aos.h:451-458
template <class F> F dot(const Vec<F>& a, const Vec<F>& b) {
    F r = a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
template <class F> F length_sqr(const Vec<F>& a) { return dot(a, a); }            // vec_aos.h:460-467
template <class F> F length(const Vec<F>& a) { return f_sqrt(length_sqr(a)); }    // vec_aos.h:469-472
template <class F> Vec<F> normalize(const Vec<F>& a) {                             // vec_aos.h:474-485
    F len_inv = F(1.0f) / f_sqrt(length_sqr(a));
    return a * len_inv;
}
template <class F> Vec<F> cross(const Vec<F>& a, const Vec<F>& b) {               // vec_aos.h:487-494
    return Vec<F>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// quat_aos.h:400-412
template <class F> Vec<F> rotate(F qx, F qy, F qz, F qw, const Vec<F>& v) {
    F tx = ((qw * v.x) + (qy * v.z)) - (qz * v.y);
    F ty = ((qw * v.y) + (qz * v.x)) - (qx * v.z);
    F tz = ((qw * v.z) + (qx * v.y)) - (qy * v.x);
    F tw = ((qx * v.x) + (qy * v.y)) + (qz * v.z);
    return Vec<F>((((tw * qx) + (tx * qw)) - (ty * qz)) + (tz * qy),
                  (((tw * qy) + (ty * qw)) - (tz * qx)) + (tx * qz),
                  (((tw * qz) + (tz * qw)) - (tx * qy)) + (ty * qx));
}

// ---------------------------------------------------------------------------
// optional bounding-volume hierarchy (the reference has none: ShapeList::hit is O(N))
// ---------------------------------------------------------------------------
// Only the order in which primitives are visited changes.  Which hit wins does not: a
// candidate's own t never depends on that order, and among candidates that tie for the
// minimum the scan of ShapeList.cpp:5-17 keeps the LAST Rect in list order if any Rect ties
// (Rect.cpp:17 accepts t == t1) and otherwise the FIRST Sphere (Sphere.cpp:15 needs t < t1);
// world_hit_bvh applies that rule explicitly.  Built by this file alone (median split over
// item centroids), not by the renderer's builder.
struct BvhItem {  // a bare primitive, or a run of primitives under one transform chain
    uint32_t prim_begin, prim_end;
    double lo[3], hi[3];  // world-space bounds, padded
};
struct BvhNodeCpu {
    float lo[3], hi[3];
    int left = -1, right = -1;  // children, or
    int first = 0, count = 0;   // leaf: items [first, first + count)
};
struct Bvh {
    std::vector<BvhItem> items;
    std::vector<BvhNodeCpu> nodes;
    // A tree handed in from outside (restate_set_external_tree: the renderer's own builder, exported
    // by tests/tree_export.cpp) is walked the way the GPU kernel walks it: both children of a node
    // tested on arrival, the nearer one first, the farther one deferred; the last n_big items lie
    // outside the tree and are tested first by every ray.
    bool two_child_order = false;
    int n_big = 0;
};

void quat_rotate_d(const double q[4], double v[3]) {  // quat_aos.h:400-412 in double
    const double tx = q[3] * v[0] + q[1] * v[2] - q[2] * v[1], ty = q[3] * v[1] + q[2] * v[0] - q[0] * v[2];
    const double tz = q[3] * v[2] + q[0] * v[1] - q[1] * v[0], tw = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    const double r[3] = {tw * q[0] + tx * q[3] - ty * q[2] + tz * q[1], tw * q[1] + ty * q[3] - tz * q[0] + tx * q[2],
                         tw * q[2] + tz * q[3] - tx * q[1] + ty * q[0]};
    v[0] = r[0]; v[1] = r[1]; v[2] = r[2];
}

void prim_local_bounds(const RtxScene& sc, uint32_t p, double lo[3], double hi[3]) {
    const double a = sc.prim_a[p], b = sc.prim_b[p], c = sc.prim_c[p], d = sc.prim_d[p], e = sc.prim_e[p];
    switch (sc.prim_kind[p]) {
        case RTX_PRIM_SPHERE: {
            const double r = std::fabs(d);
            lo[0] = a - r; lo[1] = b - r; lo[2] = c - r; hi[0] = a + r; hi[1] = b + r; hi[2] = c + r;
            break;
        }
        case RTX_PRIM_RECT_XY: lo[0] = a; hi[0] = b; lo[1] = c; hi[1] = d; lo[2] = hi[2] = e; break;
        case RTX_PRIM_RECT_XZ: lo[0] = a; hi[0] = b; lo[2] = c; hi[2] = d; lo[1] = hi[1] = e; break;
        default: lo[1] = a; hi[1] = b; lo[2] = c; hi[2] = d; lo[0] = hi[0] = e; break;
    }
}

int build_node(Bvh& bvh, int begin, int end) {
    BvhNodeCpu node;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, clo[3] = {1e300, 1e300, 1e300},
           chi[3] = {-1e300, -1e300, -1e300};
    for (int k = begin; k < end; ++k)
        for (int a = 0; a < 3; ++a) {
            const BvhItem& it = bvh.items[k];
            lo[a] = std::min(lo[a], it.lo[a]);
            hi[a] = std::max(hi[a], it.hi[a]);
            const double c = 0.5 * (it.lo[a] + it.hi[a]);
            clo[a] = std::min(clo[a], c);
            chi[a] = std::max(chi[a], c);
        }
    for (int a = 0; a < 3; ++a) {  // outward rounding to float
        node.lo[a] = std::nextafterf((float)lo[a], -INFINITY);
        node.hi[a] = std::nextafterf((float)hi[a], INFINITY);
    }
    const int index = (int)bvh.nodes.size();
    bvh.nodes.push_back(node);
    if (end - begin <= 2) {
        bvh.nodes[index].first = begin;
        bvh.nodes[index].count = end - begin;
        return index;
    }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
    const int mid = (begin + end) / 2;
    std::nth_element(bvh.items.begin() + begin, bvh.items.begin() + mid, bvh.items.begin() + end,
                     [axis](const BvhItem& x, const BvhItem& y) { return x.lo[axis] + x.hi[axis] < y.lo[axis] + y.hi[axis]; });
    const int l = build_node(bvh, begin, mid);
    const int r = build_node(bvh, mid, end);
    bvh.nodes[index].left = l;
    bvh.nodes[index].right = r;
    return index;
}

Bvh build_bvh(const RtxScene& sc) {
    Bvh bvh;
    uint32_t p = 0;
    while (p < sc.n_prims) {
        const uint32_t chain = sc.prim_chain[p];
        const uint32_t sb = sc.chain_begin[chain], se = sc.chain_begin[chain + 1];
        BvhItem it;
        it.prim_begin = p;
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        do {  // an untransformed primitive is an item of its own, a transformed run stays together
            double l[3], h[3];
            prim_local_bounds(sc, p, l, h);
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
            ++p;
        } while (se > sb && p < sc.n_prims && sc.prim_chain[p] == chain);
        it.prim_end = p;
        for (int a = 0; a < 3; ++a) { it.lo[a] = 1e300; it.hi[a] = -1e300; }
        for (int corner = 0; corner < 8; ++corner) {  // local corners out through the wrappers, innermost first
            double v[3] = {corner & 1 ? hi[0] : lo[0], corner & 2 ? hi[1] : lo[1], corner & 4 ? hi[2] : lo[2]};
            for (uint32_t s = se; s > sb; --s) {
                const uint32_t k = s - 1;
                if (sc.step_kind[k] == RTX_STEP_TRANSLATE) {
                    v[0] += sc.step_x[k]; v[1] += sc.step_y[k]; v[2] += sc.step_z[k];
                } else {
                    const double q[4] = {sc.step_x[k], sc.step_y[k], sc.step_z[k], sc.step_w[k]};
                    quat_rotate_d(q, v);
                }
            }
            for (int a = 0; a < 3; ++a) { it.lo[a] = std::min(it.lo[a], v[a]); it.hi[a] = std::max(it.hi[a], v[a]); }
        }
        double mag = 0.0;
        for (int a = 0; a < 3; ++a) mag = std::max(mag, std::max(std::fabs(it.lo[a]), std::fabs(it.hi[a])));
        for (int a = 0; a < 3; ++a) {  // fp32 slack of the transformed tests
            it.lo[a] -= 1e-3 * mag + 1e-4;
            it.hi[a] += 1e-3 * mag + 1e-4;
        }
        bvh.items.push_back(it);
    }
    if (!bvh.items.empty()) build_node(bvh, 0, (int)bvh.items.size());
    return bvh;
}

template <class F>
struct Hit {  // HitRec.h:3-10
    F t;
    Vec<F> p, n;
    int prim = -1;
    uint32_t material = 0;
    float u = 0.0f, v = 0.0f;  // Rect.cpp:27-28, Sphere.cpp:21; never counted as flops (no
                               // texture of the counted scenes reads them)
};

// ---------------------------------------------------------------------------
// the tracer
// ---------------------------------------------------------------------------
template <class F>
class Tracer {
public:
    Tracer(const RtxScene& s, int max_depth, const Bvh* bvh = nullptr) : sc(s), m_max_depth(max_depth), m_bvh(bvh) {}
    RtxRng rng{0u, 1u};
    unsigned long long segments = 0;
    unsigned long long tree_nodes = 0;  // inner nodes visited by world_hit_tree

    F draw() { return F(rtx_rng_nextf(&rng)); }  // drand48(), Util.h:10-12

    // Rect::hit, Rect.cpp:6-34 (t only; the record is completed by the caller)
    bool rect_hit(uint32_t axis, F x0, F x1, F y0, F y1, F k, const Vec<F>& o, const Vec<F>& d, F t0, F t1, F& t_out) const {
        int xi, yi, zi;
        switch (axis) {
            case RTX_PRIM_RECT_XY: xi = 0; yi = 1; zi = 2; break;
            case RTX_PRIM_RECT_XZ: xi = 0; yi = 2; zi = 1; break;
            default: xi = 1; yi = 2; zi = 0; break;
        }
        F t = (k - o.at(zi)) / d.at(zi);
        if (t < t0 || t > t1) return false;
        F x = o.at(xi) + t * d.at(xi);
        F y = o.at(yi) + t * d.at(yi);
        if (x < x0 || x > x1 || y < y0 || y > y1) return false;
        t_out = t;
        return true;
    }

    // Sphere::hit, Sphere.cpp:7-36 (t only)
    bool sphere_hit(const Vec<F>& c, F radius, const Vec<F>& o, const Vec<F>& d, F t0, F t1, F& t_out) const {
        Vec<F> oc = o - c;
        F a = dot(d, d);
        F b = F(2.0f) * dot(oc, d);
        F cc = dot(oc, oc) - radius * radius;
        F D = b * b - F(4.0f) * a * cc;
        if (D > F(0.0f)) {
            F root = f_sqrt(D);
            F temp = (-b - root) / (F(2.0f) * a);
            if (temp < t1 && temp > t0) { t_out = temp; return true; }
            temp = (-b + root) / (F(2.0f) * a);
            if (temp < t1 && temp > t0) { t_out = temp; return true; }
        }
        return false;
    }

    // ShapeList::hit (ShapeList.cpp:5-17) over the flattened leaves.  Leaves that share a
    // transform chain sit under the same Translate / Rotate wrappers in the reference, so
    // the ray is transformed once per run (Translate.cpp:7, Rotate.cpp:7-9) and a hit found
    // inside the run is carried back out once (Rotate.cpp:12-13, Translate.cpp:9).
    bool world_hit(const Vec<F>& o, const Vec<F>& d, F t0, F t1, Hit<F>& rec) {
        ++segments;
        if (m_bvh && m_bvh->two_child_order) return world_hit_tree(o, d, t0, t1, rec);
        if (m_bvh) return world_hit_bvh(o, d, t0, t1, rec);
        F closest = t1;
        bool any = false;
        uint32_t p = 0;
        while (p < sc.n_prims) {
            const uint32_t chain = sc.prim_chain[p];
            const uint32_t sb = sc.chain_begin[chain], se = sc.chain_begin[chain + 1];
            Vec<F> ro = o, rd = d;
            for (uint32_t s = sb; s < se; ++s) {
                if (sc.step_kind[s] == RTX_STEP_TRANSLATE) {
                    ro = ro - Vec<F>(sc.step_x[s], sc.step_y[s], sc.step_z[s]);
                } else {
                    F qx = -sc.step_x[s], qy = -sc.step_y[s], qz = -sc.step_z[s], qw = sc.step_w[s];  // conj
                    ro = rotate(qx, qy, qz, qw, ro);
                    rd = rotate(qx, qy, qz, qw, rd);
                }
            }
            bool run_hit = false;
            Hit<F> local;
            for (; p < sc.n_prims && sc.prim_chain[p] == chain; ++p) {
                F t;
                const uint32_t kind = sc.prim_kind[p];
                bool ok;
                if (kind == RTX_PRIM_SPHERE)
                    ok = sphere_hit(Vec<F>(sc.prim_a[p], sc.prim_b[p], sc.prim_c[p]), F(sc.prim_d[p]), ro, rd, t0, closest, t);
                else
                    ok = rect_hit(kind, sc.prim_a[p], sc.prim_b[p], sc.prim_c[p], sc.prim_d[p], sc.prim_e[p], ro, rd, t0, closest, t);
                if (!ok) continue;
                closest = t;
                run_hit = true;
                local.t = t;
                local.p = ro + rd * t;  // Ray::at, Ray.h:12
                if (kind == RTX_PRIM_SPHERE) {
                    local.n = (local.p - Vec<F>(sc.prim_a[p], sc.prim_b[p], sc.prim_c[p])) / F(sc.prim_d[p]);  // Sphere.cpp:19
                } else {
                    local.n = Vec<F>(kind == RTX_PRIM_RECT_YZ ? 1.0f : 0.0f, kind == RTX_PRIM_RECT_XZ ? 1.0f : 0.0f,
                                     kind == RTX_PRIM_RECT_XY ? 1.0f : 0.0f);  // Rect.cpp:11-13,32
                }
                fill_uv(p, kind, ro, rd, t, local);
                if (sc.prim_flip[p]) local.n = -local.n;  // FlipNormals.cpp:8
                local.prim = (int)p;
                local.material = sc.prim_material[p];
            }
            if (run_hit) {
                for (uint32_t s = se; s > sb; --s) {
                    const uint32_t k = s - 1;
                    if (sc.step_kind[k] == RTX_STEP_TRANSLATE) {
                        local.p = local.p + Vec<F>(sc.step_x[k], sc.step_y[k], sc.step_z[k]);
                    } else {
                        F qx = sc.step_x[k], qy = sc.step_y[k], qz = sc.step_z[k], qw = sc.step_w[k];
                        local.p = rotate(qx, qy, qz, qw, local.p);
                        local.n = rotate(qx, qy, qz, qw, local.n);
                    }
                }
                rec = local;
                any = true;
            }
        }
        return any;
    }

    // The same nearest hit through the hierarchy.  Slab test per node: 6 subtractions and 6
    // multiplications (counted), min / max (not counted, like every comparison).
    bool world_hit_bvh(const Vec<F>& o, const Vec<F>& d, F t0, F t1, Hit<F>& rec) {
        const Vec<F> inv(F(1.0f) / d.x, F(1.0f) / d.y, F(1.0f) / d.z);
        F closest = t1;
        bool any = false, best_is_rect = false;
        int stack[64], sp = 0;
        if (m_bvh->nodes.empty()) return false;
        stack[sp++] = 0;
        while (sp > 0) {
            const BvhNodeCpu& N = m_bvh->nodes[stack[--sp]];
            float tn = raw(t0), tf = raw(closest);
            bool miss = false;
            for (int a = 0; a < 3; ++a) {
                const float ta = raw((F(N.lo[a]) - o.at(a)) * inv.at(a)), tb = raw((F(N.hi[a]) - o.at(a)) * inv.at(a));
                if (ta != ta || tb != tb) continue;  // 0 * inf: the ray lies in a bounding plane
                tn = std::max(tn, std::min(ta, tb));
                tf = std::min(tf, std::max(ta, tb));
                if (tn > tf) { miss = true; break; }
            }
            if (miss) continue;
            if (N.count == 0) {
                if (sp + 2 > 64) return any;  // cannot happen: median splits keep the depth at log2(items)
                stack[sp++] = N.left;
                stack[sp++] = N.right;
                continue;
            }
            for (int k = N.first; k < N.first + N.count; ++k)
                test_item(m_bvh->items[k], o, d, t0, t1, closest, any, best_is_rect, rec);
        }
        return any;
    }

    // One item of a hierarchy (a bare primitive, or a run of primitives under one transform chain)
    // against the ray: the body of ShapeList::hit's loop for those primitives, with ties between
    // equal distances resolved by the list-order rule stated above the hierarchy.
    void test_item(const BvhItem& it, const Vec<F>& o, const Vec<F>& d, F t0, F t1, F& closest, bool& any, bool& best_is_rect,
                   Hit<F>& rec) {
        const uint32_t chain = sc.prim_chain[it.prim_begin];
        const uint32_t sb = sc.chain_begin[chain], se = sc.chain_begin[chain + 1];
        Vec<F> ro = o, rd = d;
        for (uint32_t s = sb; s < se; ++s) {  // Translate.cpp:7, Rotate.cpp:7-9
            if (sc.step_kind[s] == RTX_STEP_TRANSLATE) {
                ro = ro - Vec<F>(sc.step_x[s], sc.step_y[s], sc.step_z[s]);
            } else {
                F qx = -sc.step_x[s], qy = -sc.step_y[s], qz = -sc.step_z[s], qw = sc.step_w[s];
                ro = rotate(qx, qy, qz, qw, ro);
                rd = rotate(qx, qy, qz, qw, rd);
            }
        }
        for (uint32_t p = it.prim_begin; p < it.prim_end; ++p) {
            const uint32_t kind = sc.prim_kind[p];
            const bool is_rect = kind != RTX_PRIM_SPHERE;
            F t;
            bool ok;
            // the primitive's own candidate: the upper end of the window is open for a
            // sphere and closed for a rect, exactly as when it is the scan's current t1
            if (is_rect)
                ok = rect_hit(kind, sc.prim_a[p], sc.prim_b[p], sc.prim_c[p], sc.prim_d[p], sc.prim_e[p], ro, rd, t0, closest, t);
            else
                ok = sphere_hit(Vec<F>(sc.prim_a[p], sc.prim_b[p], sc.prim_c[p]), F(sc.prim_d[p]), ro, rd, t0, t1, t) &&
                     !(t > closest);
            if (!ok) continue;
            bool take = !any || t < closest;
            if (!take) {  // tie with the current winner
                const uint32_t cur = (uint32_t)rec.prim;
                if (is_rect && best_is_rect) take = p > cur;
                else if (is_rect != best_is_rect) take = is_rect;
                else take = p < cur;
            }
            if (!take) continue;
            closest = t;
            any = true;
            best_is_rect = is_rect;
            Hit<F> local;
            local.t = t;
            local.p = ro + rd * t;
            if (!is_rect)
                local.n = (local.p - Vec<F>(sc.prim_a[p], sc.prim_b[p], sc.prim_c[p])) / F(sc.prim_d[p]);
            else
                local.n = Vec<F>(kind == RTX_PRIM_RECT_YZ ? 1.0f : 0.0f, kind == RTX_PRIM_RECT_XZ ? 1.0f : 0.0f,
                                 kind == RTX_PRIM_RECT_XY ? 1.0f : 0.0f);
            fill_uv(p, kind, ro, rd, t, local);
            if (sc.prim_flip[p]) local.n = -local.n;
            local.prim = (int)p;
            local.material = sc.prim_material[p];
            for (uint32_t s = se; s > sb; --s) {  // Rotate.cpp:12-13, Translate.cpp:9
                const uint32_t q = s - 1;
                if (sc.step_kind[q] == RTX_STEP_TRANSLATE) {
                    local.p = local.p + Vec<F>(sc.step_x[q], sc.step_y[q], sc.step_z[q]);
                } else {
                    F qx = sc.step_x[q], qy = sc.step_y[q], qz = sc.step_z[q], qw = sc.step_w[q];
                    local.p = rotate(qx, qy, qz, qw, local.p);
                    local.n = rotate(qx, qy, qz, qw, local.n);
                }
            }
            rec = local;
        }
    }

    // The nearest hit through a tree handed in from outside, walked as the GPU kernel walks its
    // two-child nodes (trace_bvh_kernel in raytrace_amd/csrc/rtx_core.h): the items outside the tree
    // first; then from the root, whose own box is not tested: at an inner node both children's boxes
    // against the ray (12 counted operations per box: one subtraction and one multiplication per
    // plane), the nearer child next and the farther one deferred when both are met; a leaf tests its
    // item.  A deferred child is not tested again when it is taken up.
    bool world_hit_tree(const Vec<F>& o, const Vec<F>& d, F t0, F t1, Hit<F>& rec) {
        const Vec<F> inv(F(1.0f) / d.x, F(1.0f) / d.y, F(1.0f) / d.z);
        F closest = t1;
        bool any = false, best_is_rect = false;
        const int n_items = (int)m_bvh->items.size();
        for (int k = n_items - m_bvh->n_big; k < n_items; ++k) test_item(m_bvh->items[k], o, d, t0, t1, closest, any, best_is_rect, rec);
        if (m_bvh->nodes.empty()) return any;
        auto slab = [&](const BvhNodeCpu& N, float& t_near) {
            float tn = raw(t0), tf = raw(closest);
            for (int a = 0; a < 3; ++a) {
                const float ta = raw((F(N.lo[a]) - o.at(a)) * inv.at(a)), tb = raw((F(N.hi[a]) - o.at(a)) * inv.at(a));
                if (ta != ta || tb != tb) continue;  // 0 * inf: the ray lies in a bounding plane
                tn = std::max(tn, std::min(ta, tb));
                tf = std::min(tf, std::max(ta, tb));
            }
            t_near = tn;
            return tn <= tf;
        };
        int stack[64], sp = 0, cur = 0;
        for (;;) {
            const BvhNodeCpu& N = m_bvh->nodes[cur];
            if (N.left < 0) {  // leaf
                for (int k = N.first; k < N.first + N.count; ++k) test_item(m_bvh->items[k], o, d, t0, t1, closest, any, best_is_rect, rec);
                if (sp == 0) break;
                cur = stack[--sp];
                continue;
            }
            ++tree_nodes;
            float n0, n1;
            const bool h0 = slab(m_bvh->nodes[N.left], n0), h1 = slab(m_bvh->nodes[N.right], n1);
            if (h0 && h1) {
                const bool first0 = n0 <= n1;
                if (sp >= 64) return any;  // (the builder bounds the depth far below this)
                stack[sp++] = first0 ? N.right : N.left;
                cur = first0 ? N.left : N.right;
            } else if (h0 || h1) {
                cur = h0 ? N.left : N.right;
            } else {
                if (sp == 0) break;
                cur = stack[--sp];
            }
        }
        return any;
    }

    // hrec.u / hrec.v in the primitive's frame: Rect.cpp:21-28, Sphere.cpp:21 + Util.h:78-83
    void fill_uv(uint32_t p, uint32_t kind, const Vec<F>& ro, const Vec<F>& rd, F t, Hit<F>& h) const {
        if (kind == RTX_PRIM_SPHERE) {
            const float nx = raw(h.n.x), ny = raw(h.n.y), nz = raw(h.n.z);
            float phi = rtx_atan2f(nz, nx);
            float theta = rtx_asinf_with_root(ny, ::sqrtf((1.0f - ny) * (1.0f + ny)));
            h.u = 1.f - (phi + kPI) / (2.f * kPI);
            h.v = (theta + kPI / 2.f) / kPI;
        } else {
            const int xi = kind == RTX_PRIM_RECT_YZ ? 1 : 0, yi = kind == RTX_PRIM_RECT_XY ? 1 : 2;
            float x = raw(ro.at(xi)) + raw(t) * raw(rd.at(xi));
            float y = raw(ro.at(yi)) + raw(t) * raw(rd.at(yi));
            h.u = (x - sc.prim_a[p]) / (sc.prim_b[p] - sc.prim_a[p]);
            h.v = (y - sc.prim_c[p]) / (sc.prim_d[p] - sc.prim_c[p]);
        }
    }

    // Texture::value: ColorTexture.h:11-13, CheckerTexture.cpp:3-11, ImageTexture.h:16-29
    Vec<F> texture_value(uint32_t tex, const Hit<F>& h) const {
        if (sc.tex_kind[tex] == RTX_TEX_COLOR) return Vec<F>(sc.tex_r[tex], sc.tex_g[tex], sc.tex_b[tex]);
        if (sc.tex_kind[tex] == RTX_TEX_IMAGE) {
            const int w = (int)sc.tex_odd[tex], hgt = (int)sc.tex_even[tex];
            int i = (h.u) * w;
            int j = (1 - h.v) * hgt - 0.001;
            i = i < 0 ? 0 : i >= w ? w - 1 : i;
            j = j < 0 ? 0 : j >= hgt ? hgt - 1 : j;
            const uint8_t* t = sc.texels + sc.tex_offset[tex] + 3 * i + 3 * w * j;
            return Vec<F>(F((float)(int(t[0]) / 255.0)), F((float)(int(t[1]) / 255.0)), F((float)(int(t[2]) / 255.0)));
        }
        if (sc.tex_kind[tex] == RTX_TEX_SKY)  // Scene::backgroundSky (Scene.h:24-28) of p - centre
            return background_sky(h.p - Vec<F>(sc.tex_r[tex], sc.tex_g[tex], sc.tex_b[tex]));
        const Vec<F>& p = h.p;
        F freq = sc.tex_r[tex];
        F sines = f_sin(freq * p.x) * f_sin(freq * p.y) * f_sin(freq * p.z);
        return texture_value(sines < F(0.0f) ? sc.tex_odd[tex] : sc.tex_even[tex], h);
    }

    struct Basis { Vec<F> u, v, w; };
    // ONB::build_from_w, ONB.h:27-38
    Basis basis_from_w(const Vec<F>& n) const {
        Basis b;
        b.w = normalize(n);
        Vec<F> a = (f_abs(b.w.x) > F(0.9f)) ? Vec<F>(0.0f, 1.0f, 0.0f) : Vec<F>(1.0f, 0.0f, 0.0f);
        b.v = normalize(cross(b.w, a));
        b.u = cross(b.w, b.v);
        return b;
    }
    // ONB::local, ONB.h:24-26
    Vec<F> basis_local(const Basis& b, const Vec<F>& a) const { return b.u * a.x + b.v * a.y + b.w * a.z; }

    // A light entry's frame (general light sampling only; the reference's Translate / Rotate keep
    // Shape's default pdf_value / random, Shape.h:8-9): the query enters it as a ray enters
    // Translate::hit / Rotate::hit (Translate.cpp:7, Rotate.cpp:7-9).
    bool general_lights() const { return sc.light_sampling == RTX_LIGHTS_GENERAL; }
    void light_chain_of(uint32_t l, uint32_t& sb, uint32_t& se) const {
        const uint32_t c = general_lights() && sc.light_chain ? sc.light_chain[l] : 0u;
        sb = sc.chain_begin[c];
        se = sc.chain_begin[c + 1];
    }
    void enter_light_frame(uint32_t l, Vec<F>& o, Vec<F>* v) const {
        uint32_t sb, se;
        light_chain_of(l, sb, se);
        for (uint32_t s = sb; s < se; ++s) {
            if (sc.step_kind[s] == RTX_STEP_TRANSLATE) {
                o = o - Vec<F>(sc.step_x[s], sc.step_y[s], sc.step_z[s]);
            } else {
                F qx = -sc.step_x[s], qy = -sc.step_y[s], qz = -sc.step_z[s], qw = sc.step_w[s];  // conj
                o = rotate(qx, qy, qz, qw, o);
                if (v) *v = rotate(qx, qy, qz, qw, *v);
            }
        }
    }
    bool samples_as_rect(uint32_t kind) const {
        return kind == RTX_LIGHT_RECT_XZ || (general_lights() && kind <= RTX_LIGHT_RECT_YZ);
    }

    // Shape::pdf_value of one light entry: ShapeList.cpp:19-26 (a nested list), Rect.cpp:36-48 (the
    // XZ rect; with general light sampling the same for the rect's own axis), Sphere.cpp:38-50, Shape.h:8
    F light_entry_pdf(uint32_t l, const Vec<F>& o_in, const Vec<F>& v_in) const {
        const uint32_t kind = sc.light_kind[l];
        if (kind == RTX_LIGHT_LIST) {
            const uint32_t first = sc.light_first[l], n = sc.light_count[l];
            F weight = F(1.0f) / F((float)n);
            F sum = 0.0f;
            for (uint32_t k = 0; k < n; ++k) sum = sum + weight * light_entry_pdf(first + k, o_in, v_in);
            return sum;
        }
        Vec<F> o = o_in, v = v_in;
        enter_light_frame(l, o, &v);
        if (samples_as_rect(kind)) {
            F t;
            if (!rect_hit(kind, sc.light_a[l], sc.light_b[l], sc.light_c[l], sc.light_d[l], sc.light_e[l], o, v,
                          F(0.001f), F(FLT_MAX), t))
                return F(0.0f);
            F area = (F(sc.light_b[l]) - F(sc.light_a[l])) * (F(sc.light_d[l]) - F(sc.light_c[l]));
            F distance_squared = (t * t) * length_sqr(v);
            const Vec<F> normal(kind == RTX_LIGHT_RECT_YZ ? 1.0f : 0.0f, kind == RTX_LIGHT_RECT_XZ ? 1.0f : 0.0f,
                                kind == RTX_LIGHT_RECT_XY ? 1.0f : 0.0f);
            F cosine = f_abs(dot(v, normal)) / length(v);
            return distance_squared / (cosine * area);
        }
        if (kind == RTX_LIGHT_SPHERE) {
            F t;
            Vec<F> c(sc.light_a[l], sc.light_b[l], sc.light_c[l]);
            F radius = sc.light_d[l];
            if (!sphere_hit(c, radius, o, v, F(0.001f), F(FLT_MAX), t)) return F(0.0f);
            F dd = length_sqr(c - o);
            F r2 = radius * radius;
            F rr = dd < r2 ? dd : r2;  // std::min(pow2(m_radius), dd)
            F cos_theta_max = f_sqrt(F(1.0f) - rr * (F(1.0f) / dd));
            F solid_angle = F(kPI2) * (F(1.0f) - cos_theta_max);
            return F(1.0f) / solid_angle;
        }
        return F(0.0f);
    }

    // ShapePdf::value (ShapePdf.h:11-13) -> ShapeList::pdf_value (ShapeList.cpp:19-26)
    F light_pdf(const Vec<F>& o, const Vec<F>& v) const {
        if (!sc.light_is_list) return light_entry_pdf(0, o, v);
        F weight = F(1.0f) / F((float)sc.n_lights);
        F sum = 0.0f;
        for (uint32_t l = 0; l < sc.n_lights; ++l) sum = sum + weight * light_entry_pdf(l, o, v);
        return sum;
    }

    // Shape::random of one light entry: ShapeList.cpp:28-35 (a nested list draws again), Rect.cpp:50-68,
    // Sphere.cpp:52-58 (+ Util.h:96-107), Shape.h:9
    Vec<F> light_entry_random(uint32_t l, const Vec<F>& o_in) {
        const uint32_t kind = sc.light_kind[l];
        if (kind == RTX_LIGHT_LIST) {
            size_t n = sc.light_count[l];
            size_t index = size_t(raw(draw()) * (float)n);
            if (n > 0 && index >= n) index = n - 1;
            return light_entry_random(sc.light_first[l] + (uint32_t)index, o_in);
        }
        if (!samples_as_rect(kind) && kind != RTX_LIGHT_SPHERE) return Vec<F>(1.0f, 0.0f, 0.0f);
        Vec<F> o = o_in;
        enter_light_frame(l, o, nullptr);
        Vec<F> dir;
        if (samples_as_rect(kind)) {
            F x = F(sc.light_a[l]) + draw() * (F(sc.light_b[l]) - F(sc.light_a[l]));
            F y = F(sc.light_c[l]) + draw() * (F(sc.light_d[l]) - F(sc.light_c[l]));
            const F k = sc.light_e[l];
            const Vec<F> point = kind == RTX_LIGHT_RECT_XY ? Vec<F>(x, y, k) : (kind == RTX_LIGHT_RECT_XZ ? Vec<F>(x, k, y) : Vec<F>(k, x, y));
            dir = point - o;
        } else {
            Vec<F> direction = Vec<F>(sc.light_a[l], sc.light_b[l], sc.light_c[l]) - o;
            F distance_squared = length_sqr(direction);
            Basis uvw = basis_from_w(direction);
            F radius = sc.light_d[l];
            F r1 = draw();
            F r2 = draw();
            F rad2 = radius * radius;
            F rr = distance_squared < rad2 ? distance_squared : rad2;
            F cos_theta_max = f_sqrt(F(1.0f) - rr * (F(1.0f) / distance_squared));
            F z = F(1.0f) - r2 * (F(1.0f) - cos_theta_max);
            F sqrtz = f_sqrt(F(1.0f) - z * z);
            F phi = F(kPI2) * r1;
            F sn, cs;
            f_sincos(phi, sn, cs);
            dir = basis_local(uvw, Vec<F>(cs * sqrtz, sn * sqrtz, z));
        }
        // out of the entry's frame as a normal leaves Rotate::hit (Rotate.cpp:13), innermost wrapper first
        uint32_t sb, se;
        light_chain_of(l, sb, se);
        for (uint32_t s = se; s > sb; --s) {
            const uint32_t q = s - 1;
            if (sc.step_kind[q] != RTX_STEP_ROTATE) continue;
            F qx = sc.step_x[q], qy = sc.step_y[q], qz = sc.step_z[q], qw = sc.step_w[q];
            dir = rotate(qx, qy, qz, qw, dir);
        }
        return dir;
    }

    // ShapePdf::generate (ShapePdf.h:15-17) -> ShapeList::random (ShapeList.cpp:28-35)
    Vec<F> light_random(const Vec<F>& o) {
        if (!sc.light_is_list) return light_entry_random(0, o);
        size_t n = sc.n_lights;
        size_t index = size_t(raw(draw()) * (float)n);
        if (n > 0 && index >= n) index = n - 1;
        return light_entry_random((uint32_t)index, o);
    }

    // Scene::background (Scene.h:20-22) or Scene::backgroundSky (Scene.h:24-28; lerp, vec_aos.h:92-95)
    Vec<F> background(const Vec<F>& d) const {
        if (sc.background_mode != RTX_BACKGROUND_SKY) return Vec<F>(sc.background[0], sc.background[1], sc.background[2]);
        return background_sky(d);
    }
    Vec<F> background_sky(const Vec<F>& d) const {  // Scene.h:24-28
        Vec<F> v = normalize(d);
        F t = F(0.5f) * (v.y + F(1.0f));
        Vec<F> a(1.0f, 1.0f, 1.0f), b(0.5f, 0.7f, 1.0f);
        return a + (b - a) * t;
    }

    // CosinePdf::generate (CosinePdf.cpp:16-20) with random_cosine_direction (Util.h:86-94)
    Vec<F> cosine_generate(const Vec<F>& n) {
        Basis uvw = basis_from_w(n);
        F r1 = draw();
        F r2 = draw();
        F z = f_sqrt(F(1.0f) - r2);
        F phi = F(kPI2) * r1;
        F sn, cs;
        f_sincos(phi, sn, cs);
        F x = cs * f_sqrt(r2);
        F y = sn * f_sqrt(r2);
        return basis_local(uvw, Vec<F>(x, y, z));
    }

    // CosinePdf::value, CosinePdf.cpp:6-14
    F cosine_value(const Vec<F>& n, const Vec<F>& dir) const {
        F cosine = dot(normalize(dir), n);
        return cosine > F(0.0f) ? cosine / F(kPI) : F(0.0f);
    }

    // Material::scatter (Material.h:9) of the four materials: what ScatterRec (ScatterRec.h:7-12)
    // carries out of it.  `albedo` is the value of the material's texture at the hit.
    struct Scattered {
        bool ok;        // scatter()'s return value
        bool specular;  // ScatterRec::is_specular
        Vec<F> dir;     // ScatterRec::ray's direction (specular materials; its origin is the hit point)
        Vec<F> albedo;  // ScatterRec::albedo
    };
    Scattered scatter(uint32_t kind, F param, const Vec<F>& albedo, const Vec<F>& d, const Vec<F>& n) {
        Scattered s;
        s.ok = false;
        s.specular = false;
        s.dir = Vec<F>(0.0f, 0.0f, 0.0f);
        s.albedo = albedo;
        switch (kind) {
            case RTX_MAT_DIFFUSE_LIGHT:  // DiffuseLight.h:17-19: scatter() is false
                return s;
            case RTX_MAT_METAL: {  // Metal.cpp:9-17
                Vec<F> nd = normalize(d);
                Vec<F> reflected = nd - n * (F(2.0f) * dot(nd, n));  // reflect, Util.h:56-58
                Vec<F> q;
                do {  // random_in_unit_sphere, Util.h:32-38; g++ evaluates the arguments of
                      // Vector3(drand48(), drand48(), drand48()) (Util.h:28-30) right to left
                    F rz = draw();
                    F ry = draw();
                    F rx = draw();
                    q = Vec<F>(rx, ry, rz) * F(2.0f) - Vec<F>(1.0f, 1.0f, 1.0f);
                } while (length_sqr(q) >= F(1.0f));
                reflected = reflected + q * param;
                s.specular = true;
                s.dir = reflected;
                s.ok = dot(reflected, n) > F(0.0f);
                return s;
            }
            case RTX_MAT_DIELECTRIC: {  // Dielectric.cpp:7-45
                F ri = param;
                Vec<F> reflected = d - n * (F(2.0f) * dot(d, n));
                Vec<F> outward;
                F ni_over_nt, cosine;
                if (dot(d, n) > F(0.0f)) {
                    outward = -n;
                    ni_over_nt = ri;
                    cosine = ri * dot(d, n) / length(d);
                } else {
                    outward = n;
                    ni_over_nt = F(1.0f) / ri;
                    cosine = -dot(d, n) / length(d);
                }
                // refract(-d, outward, ni_over_nt), Util.h:60-71
                Vec<F> refracted;
                F reflect_prob = 1.0f;
                {
                    Vec<F> uv = normalize(-d);
                    F dt = dot(uv, outward);
                    F D = F(1.0f) - (ni_over_nt * ni_over_nt) * (F(1.0f) - dt * dt);
                    if (D > F(0.0f)) {
                        refracted = (uv - outward * dt) * (-ni_over_nt) - outward * f_sqrt(D);
                        // schlick, Util.h:73-76
                        F q = (F(1.0f) - ri) / (F(1.0f) + ri);
                        F r0 = q * q;
                        F mm = F(1.0f) - cosine;
                        reflect_prob = r0 + (F(1.0f) - r0) * (mm * mm * mm * mm * mm);
                    }
                }
                s.dir = (draw() < reflect_prob) ? reflected : refracted;
                s.albedo = Vec<F>(1.0f, 1.0f, 1.0f);  // Dielectric.cpp:8
                s.specular = true;
                s.ok = true;
                return s;
            }
            default:  // Lambertian.cpp:16-21: the direction is drawn by the caller from srec.pdf (CosinePdf)
                s.ok = true;
                return s;
        }
    }
    // Lambertian::scattering_pdf, Lambertian.cpp:23-25
    F scattering_pdf(const Vec<F>& n, const Vec<F>& dir) const {
        F c2 = dot(n, normalize(dir));
        return (c2 < F(0.0f) ? F(0.0f) : c2) / F(kPI);
    }
    // Material::emitted: Material.h:10, DiffuseLight.h:21-28 (`value` = the emission texture at the hit)
    Vec<F> emitted_of(uint32_t kind, const Vec<F>& value, const Vec<F>& n, const Vec<F>& d) const {
        if (kind == RTX_MAT_DIFFUSE_LIGHT && dot(n, d) < F(0.0f)) return value;
        return Vec<F>(0.0f, 0.0f, 0.0f);
    }

    // Scene::color, Scene.cpp:103-132
    Vec<F> color(const Vec<F>& o, const Vec<F>& d, int depth) {
        Hit<F> rec;
        if (!world_hit(o, d, F(0.001f), F(FLT_MAX), rec)) return background(d);
        const uint32_t m = rec.material;
        const uint32_t kind = sc.mat_kind[m];
        Vec<F> emitted(0.0f, 0.0f, 0.0f);
        if (kind == RTX_MAT_DIFFUSE_LIGHT && dot(rec.n, d) < F(0.0f)) emitted = texture_value(sc.mat_texture[m], rec);
        if (!(depth < m_max_depth)) return emitted;
        if (kind == RTX_MAT_DIFFUSE_LIGHT) return emitted;
        const Vec<F> albedo = kind == RTX_MAT_DIELECTRIC ? Vec<F>(1.0f, 1.0f, 1.0f) : Vec<F>(0.0f, 0.0f, 0.0f);
        // (the albedo texture of Metal.cpp:13 / Lambertian.cpp:17 is evaluated below, where its
        // value is used: it draws no random numbers, so the order does not show)
        Scattered s = scatter(kind, F(sc.mat_param[m]), albedo, d, rec.n);
        if (s.specular) {
            if (kind == RTX_MAT_METAL) s.albedo = texture_value(sc.mat_texture[m], rec);
            if (!s.ok) return emitted;
            return emitted + mul_per_elem(s.albedo, color(rec.p, s.dir, depth + 1));  // Scene.cpp:110
        }
        {   // Lambertian.cpp:16-25 + Scene.cpp:112-125
            s.albedo = texture_value(sc.mat_texture[m], rec);
            Vec<F> dir;
            F pdf_value;
            if (sc.n_lights == 0) {
                // no light shape (not expressible in the reference, whose color() always mixes
                // with a ShapePdf): the material's own pdf, srec.pdf of Lambertian.cpp:19, alone
                dir = cosine_generate(rec.n);
                pdf_value = cosine_value(rec.n, dir);
            } else {
                if (draw() < F(0.5f))  // MixturePdf::generate, MixturePdf.h:15-22
                    dir = light_random(rec.p);
                else
                    dir = cosine_generate(rec.n);
                // MixturePdf::value, MixturePdf.h:9-13
                F pdf0 = light_pdf(rec.p, dir);
                F pdf1 = cosine_value(rec.n, dir);
                pdf_value = F(0.5f) * pdf0 + F(0.5f) * pdf1;
            }
            if (!(pdf_value > F(0.0f))) return emitted;
            F spdf = scattering_pdf(rec.n, dir);
            Vec<F> weight = s.albedo * spdf;
            return emitted + mul_per_elem(weight, color(rec.p, dir, depth + 1)) / pdf_value;  // Scene.cpp:120
        }
    }

    // one sample of Scene::render's inner loop, Scene.cpp:145-148 (+ Camera::getRay, Camera.h:24-26)
    Vec<F> sample(int i, int j, int nx, int ny) {
        F u = (F((float)i) + draw()) / F((float)nx);
        F v = (F((float)j) + draw()) / F((float)ny);
        Vec<F> org(sc.cam_origin[0], sc.cam_origin[1], sc.cam_origin[2]);
        Vec<F> ll(sc.cam_ll[0], sc.cam_ll[1], sc.cam_ll[2]);
        Vec<F> hu(sc.cam_u[0], sc.cam_u[1], sc.cam_u[2]);
        Vec<F> hv(sc.cam_v[0], sc.cam_v[1], sc.cam_v[2]);
        Vec<F> dir = ll + hu * u + hv * v - org;
        return color(org, dir, 0);
    }

private:
    const RtxScene& sc;
    int m_max_depth;
    const Bvh* m_bvh;
};

std::atomic<int> g_use_bvh(0);
Bvh g_external_tree;                                 // restate_set_external_tree
std::atomic<int> g_has_external_tree(0);
std::atomic<unsigned long long> g_last_tree_nodes(0);  // inner nodes visited by the last render through it

template <class F>
int render_impl(const RtxScene* scene, int nx, int ny, int spp_begin, int spp_count, uint64_t seed, int max_depth,
                int n_threads, float* sum_rgb, float* per_sample_rgb, uint32_t* per_sample_segments,
                uint64_t* total_rays, uint64_t* total_flops) {
    if (!scene || nx <= 0 || ny <= 0 || spp_count < 0) return -1;
    std::atomic<int> next_row(0);
    std::atomic<unsigned long long> rays(0), flops(0);
    Bvh bvh;
    const bool external = g_has_external_tree.load() != 0;
    const bool use_bvh = g_use_bvh.load() != 0 || external;
    if (use_bvh && !external) bvh = build_bvh(*scene);
    std::atomic<unsigned long long> tree_nodes(0);
    auto worker = [&]() {
        Tracer<F> tr(*scene, max_depth, external ? &g_external_tree : (use_bvh ? &bvh : nullptr));
        g_flops = 0;
        for (;;) {
            int j = next_row.fetch_add(1);
            if (j >= ny) break;
            for (int i = 0; i < nx; ++i) {
                Vec<F> c(0.0f, 0.0f, 0.0f);
                const uint32_t pixel = uint32_t(j) * uint32_t(nx) + uint32_t(i);
                for (int s = 0; s < spp_count; ++s) {
                    rtx_rng_init(&tr.rng, pixel, uint32_t(spp_begin + s), seed);
                    unsigned long long before = tr.segments;
                    Vec<F> L = tr.sample(i, j, nx, ny);
                    c = c + L;  // Scene.cpp:148
                    const size_t k = size_t(pixel) * spp_count + s;
                    if (per_sample_rgb) {
                        per_sample_rgb[3 * k + 0] = raw(L.x);
                        per_sample_rgb[3 * k + 1] = raw(L.y);
                        per_sample_rgb[3 * k + 2] = raw(L.z);
                    }
                    if (per_sample_segments) per_sample_segments[k] = uint32_t(tr.segments - before);
                }
                if (sum_rgb) {
                    float* o = sum_rgb + (size_t(ny - j - 1) * nx + i) * 3;  // Scene.cpp:151
                    o[0] = raw(c.x);
                    o[1] = raw(c.y);
                    o[2] = raw(c.z);
                }
            }
        }
        rays += tr.segments;
        flops += g_flops;
        tree_nodes += tr.tree_nodes;
    };
    if (n_threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
    if (total_rays) *total_rays = rays.load();
    if (total_flops) *total_flops = flops.load();
    g_last_tree_nodes.store(tree_nodes.load());
    return 0;
}

}  // namespace

extern "C" {

/* Same contract as ref_render() of oracle/ref_driver.cpp, over a flattened scene. */
int restate_render(const RtxScene* scene, int nx, int ny, int spp_begin, int spp_count, uint64_t seed, int max_depth,
                   int n_threads, float* sum_rgb, float* per_sample_rgb, uint32_t* per_sample_segments,
                   uint64_t* total_rays) {
    return render_impl<float>(scene, nx, ny, spp_begin, spp_count, seed, max_depth, n_threads, sum_rgb, per_sample_rgb,
                              per_sample_segments, total_rays, nullptr);
}

/* Op-counting run: total fp32 operations and rays of the same render. */
int restate_count_flops(const RtxScene* scene, int nx, int ny, int spp_begin, int spp_count, uint64_t seed,
                        int max_depth, int n_threads, uint64_t* total_rays, uint64_t* total_flops) {
    return render_impl<Counted>(scene, nx, ny, spp_begin, spp_count, seed, max_depth, n_threads, nullptr, nullptr, nullptr,
                                total_rays, total_flops);
}

/* 1: later calls walk this file's own bounding-volume hierarchy instead of scanning the list. */
void restate_set_bvh(int on) { g_use_bvh.store(on); }

/* Later renders walk this tree instead (n_nodes = 0: back to the list / this file's own hierarchy):
 * one record per box, record 0 the root, left / right < 0 for a leaf whose items are
 * [first, first + count); the last n_big items lie outside the tree.  The arrays are copied. */
void restate_set_external_tree(int n_items, const uint32_t* item_begin, const uint32_t* item_end, int n_big, int n_nodes,
                               const float* lo, const float* hi, const int32_t* left, const int32_t* right,
                               const int32_t* first, const int32_t* count) {
    g_has_external_tree.store(0);
    g_external_tree = Bvh{};
    if (n_nodes <= 0) return;
    g_external_tree.two_child_order = true;
    g_external_tree.n_big = n_big;
    g_external_tree.items.resize((size_t)n_items);
    for (int k = 0; k < n_items; ++k) {
        g_external_tree.items[k].prim_begin = item_begin[k];
        g_external_tree.items[k].prim_end = item_end[k];
    }
    g_external_tree.nodes.resize((size_t)n_nodes);
    for (int k = 0; k < n_nodes; ++k) {
        BvhNodeCpu& N = g_external_tree.nodes[k];
        for (int a = 0; a < 3; ++a) {
            N.lo[a] = lo[3 * k + a];
            N.hi[a] = hi[3 * k + a];
        }
        N.left = left[k];
        N.right = right[k];
        N.first = first[k];
        N.count = count[k];
    }
    g_has_external_tree.store(1);
}
/* Inner nodes visited by the last render through an external tree. */
unsigned long long restate_last_tree_nodes(void) { return g_last_tree_nodes.load(); }

/* Nearest hit of n rays (Shape::hit of the world). */
int restate_hit(const RtxScene* scene, uint32_t n, const float* org, const float* dir, float t0, float t1, RtxHit* hits) {
    if (!scene) return -1;
    Bvh bvh;
    const bool use_bvh = g_use_bvh.load() != 0;
    if (use_bvh) bvh = build_bvh(*scene);
    Tracer<float> tr(*scene, 50, use_bvh ? &bvh : nullptr);
    for (uint32_t k = 0; k < n; ++k) {
        Hit<float> rec;
        Vec<float> o(org[3 * k], org[3 * k + 1], org[3 * k + 2]), d(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
        bool h = tr.world_hit(o, d, t0, t1, rec);
        RtxHit& out = hits[k];
        std::memset(&out, 0, sizeof out);
        out.hit = h ? 1 : 0;
        out.prim = h ? rec.prim : -1;
        out.material = h ? (int32_t)rec.material : -1;
        if (h) {
            out.t = rec.t;
            out.p[0] = rec.p.x; out.p[1] = rec.p.y; out.p[2] = rec.p.z;
            out.n[0] = rec.n.x; out.n[1] = rec.n.y; out.n[2] = rec.n.z;
            out.uv[0] = rec.u; out.uv[1] = rec.v;
        }
    }
    return 0;
}

/* Image::write (Image.h:30-39) with its three filters: DenanFilter.h:8-13,
 * GammaFilter.h:7-9 (Util.h:41-47), TonemapFilter.h:8-10. */
void restate_image_write(int n, const float* rgb, uint8_t* out) {
    for (int k = 0; k < n; ++k) {
        float c[3] = {rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]};
        c[0] = !(c[0] == c[0]) ? 1.0f : c[0];
        c[1] = !(c[1] == c[1]) ? 0.0f : c[1];
        c[2] = !(c[2] == c[2]) ? 1.0f : c[2];
        for (int a = 0; a < 3; ++a) {
            float g = ::powf(c[a], 1.0f / 2.2f);
            g = (g > 0.0f) ? g : 0.0f;   // maxPerElem(c, 0), vec_aos.h:418-425
            g = (g < 1.0f) ? g : 1.0f;   // minPerElem(., 1), vec_aos.h:435-442
            out[3 * k + a] = static_cast<unsigned char>(g * 255.99f);
        }
    }
}


/* The two pieces of the no-light / sky extension the reference does contain, for function-level
 * pinning: Scene::backgroundSky (Scene.h:24-28) and CosinePdf::generate / value
 * (CosinePdf.cpp:6-20), entry k drawing from the stream of (pixel k, sample 0, seed). */
int restate_background_sky(int n, const float* dirs, float* out) {
    RtxScene sc;
    std::memset(&sc, 0, sizeof sc);
    sc.background_mode = RTX_BACKGROUND_SKY;
    Tracer<float> tr(sc, 50);
    for (int k = 0; k < n; ++k) {
        Vec<float> c = tr.background(Vec<float>(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
        out[3 * k] = c.x; out[3 * k + 1] = c.y; out[3 * k + 2] = c.z;
    }
    return 0;
}

int restate_cosine_pdf(int n, const float* normals, uint64_t seed, float* dirs, float* values) {
    RtxScene sc;
    std::memset(&sc, 0, sizeof sc);
    Tracer<float> tr(sc, 50);
    for (int k = 0; k < n; ++k) {
        Vec<float> nn(normals[3 * k], normals[3 * k + 1], normals[3 * k + 2]);
        rtx_rng_init(&tr.rng, uint32_t(k), 0u, seed);
        Vec<float> d = tr.cosine_generate(nn);
        dirs[3 * k] = d.x; dirs[3 * k + 1] = d.y; dirs[3 * k + 2] = d.z;
        values[k] = tr.cosine_value(nn, d);
    }
    return 0;
}

/* ShapePdf::value / generate of the scene's light shape (ShapePdf.h:8-17 -> ShapeList.cpp:19-35,
 * Rect.cpp:36-68, Sphere.cpp:38-58, Shape.h:8-9) for n (origin, direction) pairs; entry k draws
 * from the stream of (pixel k, sample 0, seed). */
int restate_light_pdf(const RtxScene* scene, int n, const float* org, const float* dir, uint64_t seed, float* values,
                      float* dirs) {
    if (!scene || scene->n_lights == 0) return -1;
    Tracer<float> tr(*scene, 50);
    for (int k = 0; k < n; ++k) {
        const Vec<float> o(org[3 * k], org[3 * k + 1], org[3 * k + 2]), v(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
        values[k] = tr.light_pdf(o, v);
        rtx_rng_init(&tr.rng, uint32_t(k), 0u, seed);
        const Vec<float> d = tr.light_random(o);
        dirs[3 * k] = d.x; dirs[3 * k + 1] = d.y; dirs[3 * k + 2] = d.z;
    }
    return 0;
}

/* Material::scatter / ::emitted / ::scattering_pdf of one material (kind RTX_MAT_*, param = fuzz or
 * refractive index, plain-colour texture `tex`) for n (incoming direction, hit normal, scattered
 * direction) triples; entry k draws from the stream of (pixel k, sample 0, seed).  Same outputs as
 * ref_scatter() of oracle/ref_driver.cpp. */
int restate_scatter(uint32_t kind, float param, const float* tex, int n, const float* d_in, const float* normals,
                    const float* scattered, uint64_t seed, int32_t* ok, int32_t* specular, float* dir, float* albedo,
                    float* emitted, float* spdf) {
    RtxScene empty{};
    Tracer<float> tr(empty, 50, nullptr);
    const Vec<float> value(tex[0], tex[1], tex[2]);
    for (int k = 0; k < n; ++k) {
        const Vec<float> d(d_in[3 * k], d_in[3 * k + 1], d_in[3 * k + 2]), nn(normals[3 * k], normals[3 * k + 1], normals[3 * k + 2]);
        const Vec<float> sd(scattered[3 * k], scattered[3 * k + 1], scattered[3 * k + 2]);
        rtx_rng_init(&tr.rng, uint32_t(k), 0u, seed);
        auto s = tr.scatter(kind, param, value, d, nn);
        ok[k] = s.ok ? 1 : 0;
        specular[k] = s.specular ? 1 : 0;
        const Vec<float> e = tr.emitted_of(kind, value, nn, d);
        const float sp = kind == RTX_MAT_LAMBERTIAN ? tr.scattering_pdf(nn, sd) : 0.0f;  // Material.h:11: 0 by default
        dir[3 * k] = s.dir.x; dir[3 * k + 1] = s.dir.y; dir[3 * k + 2] = s.dir.z;
        albedo[3 * k] = s.albedo.x; albedo[3 * k + 1] = s.albedo.y; albedo[3 * k + 2] = s.albedo.z;
        emitted[3 * k] = e.x; emitted[3 * k + 1] = e.y; emitted[3 * k + 2] = e.z;
        spdf[k] = sp;
    }
    return 0;
}
}  // extern "C"
