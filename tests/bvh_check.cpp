// bvh_check.cpp — host-only checker of the two tree builders of raytrace_amd/csrc/rtx_bvh.cpp
// (test infrastructure: compiled with g++ by tests/test_bvh_builder.py, no GPU involved).
//
// Scene tree (rtxbvh::build): every group is in exactly one leaf or among the items kept outside
// the tree; a ray that meets a group's exact bounds reaches that group's leaf when the tree is
// walked with the decoded 16-bit boxes; the depth stays within the traversal stack.
// Light tree (rtxbvh::build_light_tree): depth-first order, skip links that point past each
// subtree, every entry in exactly one leaf, and the stackless walk finds every light whose exact
// bounds a ray meets.
//
// usage: bvh_check <n_items> <seed>; prints "ok ..." and returns 0, or a message and 1.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "rtx.h"
#include "rtx_bvh.h"

namespace {

struct Box3 {
    double lo[3], hi[3];
};

void rotate(const rtxd::Step& q, double v[3]) {  // v' = q v q*
    const double x = q.x, y = q.y, z = q.z, w = q.w;
    const double cx = y * v[2] - z * v[1], cy = z * v[0] - x * v[2], cz = x * v[1] - y * v[0];
    const double dx = y * cz - z * cy, dy = z * cx - x * cz, dz = x * cy - y * cx;
    v[0] += 2.0 * (w * cx + dx);
    v[1] += 2.0 * (w * cy + dy);
    v[2] += 2.0 * (w * cz + dz);
}

// Exact (unpadded) world bounds of one group: corners of each primitive's local box through the chain.
Box3 exact_bounds(const std::vector<rtxd::Prim>& prims, const rtxd::Group& G, const std::vector<rtxd::Step>& steps) {
    Box3 b{{1e300, 1e300, 1e300}, {-1e300, -1e300, -1e300}};
    for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) {
        const rtxd::Prim& P = prims[p];
        double lo[3], hi[3];
        const uint32_t kind = P.kind_flip & 3u;
        if (kind == RTX_PRIM_SPHERE) {
            const double c[3] = {P.a, P.b, P.c};
            for (int a = 0; a < 3; ++a) { lo[a] = c[a] - std::fabs(P.d); hi[a] = c[a] + std::fabs(P.d); }
        } else {
            const int xi = kind == RTX_PRIM_RECT_YZ ? 1 : 0, yi = kind == RTX_PRIM_RECT_XY ? 1 : 2, zi = 3 - xi - yi;
            lo[xi] = P.a; hi[xi] = P.b; lo[yi] = P.c; hi[yi] = P.d; lo[zi] = hi[zi] = P.e;
        }
        for (int c = 0; c < 8; ++c) {
            double v[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
            for (uint32_t s = G.step_end; s > G.step_begin; --s) {
                const rtxd::Step& S = steps[s - 1];
                if (S.kind == RTX_STEP_ROTATE) rotate(S, v);
                else { v[0] += S.x; v[1] += S.y; v[2] += S.z; }
            }
            for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], v[a]); b.hi[a] = std::max(b.hi[a], v[a]); }
        }
    }
    return b;
}

bool ray_box(const double o[3], const double d[3], const double lo[3], const double hi[3]) {
    double tn = 0.0, tf = 1e300;
    for (int a = 0; a < 3; ++a) {
        if (d[a] == 0.0) {
            if (o[a] < lo[a] || o[a] > hi[a]) return false;
            continue;
        }
        double t0 = (lo[a] - o[a]) / d[a], t1 = (hi[a] - o[a]) / d[a];
        if (t0 > t1) std::swap(t0, t1);
        tn = std::max(tn, t0);
        tf = std::min(tf, t1);
    }
    return tn <= tf;
}

int fail(const char* what) {
    std::printf("FAILED: %s\n", what);
    return 1;
}

}  // namespace

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 500u;
    std::mt19937 gen(argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1u);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);

    // groups: bare spheres, bare rects, and rects under a Rotate + Translate chain
    std::vector<rtxd::Prim> prims;
    std::vector<rtxd::Group> groups;
    std::vector<rtxd::Step> steps;
    for (uint32_t g = 0; g < n; ++g) {
        rtxd::Prim P{};
        rtxd::Group G{};
        G.prim_begin = (uint32_t)prims.size();
        G.step_begin = G.step_end = (uint32_t)steps.size();
        const float x = 100.0f * U(gen), y = 10.0f * U(gen), z = 100.0f * U(gen);
        const uint32_t pick = gen() % 3u;
        if (pick == 0u) {
            P.a = x; P.b = y; P.c = z; P.d = 0.05f + 0.5f * U(gen);
            P.kind_flip = RTX_PRIM_SPHERE;
        } else {
            const uint32_t kind = gen() % 3u;
            const float w = 0.1f + U(gen), h = 0.1f + U(gen);
            P.kind_flip = kind;
            if (pick == 1u) {
                const float c0 = kind == RTX_PRIM_RECT_YZ ? y : x, c1 = kind == RTX_PRIM_RECT_XY ? y : z;
                const float k = kind == RTX_PRIM_RECT_XY ? z : (kind == RTX_PRIM_RECT_XZ ? y : x);
                P.a = c0; P.b = c0 + w; P.c = c1; P.d = c1 + h; P.e = k;
            } else {
                P.a = 0.0f; P.b = w; P.c = 0.0f; P.d = h; P.e = 0.3f * U(gen);
                const float ang = 6.2831853f * U(gen);
                rtxd::Step R{};
                R.kind = RTX_STEP_ROTATE;
                R.x = 0.0f; R.y = std::sin(0.5f * ang); R.z = 0.0f; R.w = std::cos(0.5f * ang);
                rtxd::Step T{};
                T.kind = RTX_STEP_TRANSLATE;
                T.x = x; T.y = y; T.z = z;
                steps.push_back(T);  // outermost wrapper first
                steps.push_back(R);
                G.step_end = (uint32_t)steps.size();
            }
        }
        P.group = g;
        prims.push_back(P);
        G.prim_end = (uint32_t)prims.size();
        groups.push_back(G);
    }
    if (n >= 64) {  // one item far larger than the rest: must end up outside the tree
        rtxd::Prim P{};
        P.a = 50.0f; P.b = -1000.0f; P.c = 50.0f; P.d = 1000.0f;
        P.kind_flip = RTX_PRIM_SPHERE;
        P.group = (uint32_t)groups.size();
        rtxd::Group G{};
        G.prim_begin = (uint32_t)prims.size();
        G.prim_end = G.prim_begin + 1u;
        G.step_begin = G.step_end = 0u;
        prims.push_back(P);
        groups.push_back(G);
    }
    std::vector<Box3> exact(groups.size());
    for (size_t g = 0; g < groups.size(); ++g) exact[g] = exact_bounds(prims, groups[g], steps);

    // ---- scene tree ----
    rtxbvh::Tree tree;
    std::string err;
    if (!rtxbvh::build(prims, groups, steps, tree, err)) return fail(err.c_str());
    if (tree.items.size() != groups.size()) return fail("items != groups");
    if (n >= 64 && tree.n_big == 0) return fail("the oversized item is inside the tree");
    // items are groups, each exactly once
    std::vector<int> seen(groups.size(), 0);
    std::vector<uint32_t> item_group(tree.items.size());
    for (size_t k = 0; k < tree.items.size(); ++k) {
        const rtxd::Group& I = tree.items[k];
        size_t g = 0;
        for (; g < groups.size(); ++g)
            if (groups[g].prim_begin == I.prim_begin && groups[g].prim_end == I.prim_end) break;
        if (g == groups.size()) return fail("an item is not one of the groups");
        ++seen[g];
        item_group[k] = (uint32_t)g;
    }
    for (int s : seen)
        if (s != 1) return fail("a group is not in exactly one item");
    // leaves cover the in-tree items exactly once; depth bound
    const uint32_t n_tree_items = (uint32_t)tree.items.size() - tree.n_big;
    std::vector<int> leafed(n_tree_items, 0);
    struct Ref { uint32_t ref; int depth; };
    std::vector<Ref> stack{{0u, 0}};
    int max_depth = 0;
    if (tree.nodes.empty()) return fail("no nodes");
    while (!stack.empty()) {
        Ref r = stack.back();
        stack.pop_back();
        max_depth = std::max(max_depth, r.depth);
        if (r.ref & rtxd::kBvhLeaf) {
            const uint32_t first = r.ref & rtxd::kBvhLeafFirst, count = (r.ref >> rtxd::kBvhLeafCountShift) & 15u;
            for (uint32_t k = 0; k < count; ++k) {
                if (first + k >= n_tree_items) return fail("leaf item out of range");
                ++leafed[first + k];
            }
            continue;
        }
        if (r.ref >= tree.nodes.size()) return fail("node reference out of range");
        stack.push_back({tree.nodes[r.ref].child0, r.depth + 1});
        stack.push_back({tree.nodes[r.ref].child1, r.depth + 1});
    }
    // (a tree of one item is a root whose two children are both that leaf)
    for (int s : leafed)
        if (s != (n_tree_items == 1u ? 2 : 1)) return fail("an in-tree item is not in exactly one leaf");
    if (max_depth > rtxbvh::kMaxDepth) return fail("tree deeper than kMaxDepth");

    auto decode = [&](const uint16_t q[3], double out[3]) {
        for (int a = 0; a < 3; ++a) out[a] = (double)tree.origin[a] + (double)q[a] * (double)tree.scale[a];
    };
    std::uniform_real_distribution<double> D(-1.0, 1.0);
    uint32_t checked = 0;
    for (int r = 0; r < 400; ++r) {
        const double o[3] = {100.0 * U(gen), 10.0 * U(gen) + 5.0, 100.0 * U(gen)};
        double d[3] = {D(gen), D(gen), D(gen)};
        if (r % 2 == 0) {  // towards some item, so that the check has hits to look at
            const Box3& b = exact[gen() % groups.size()];
            for (int a = 0; a < 3; ++a) d[a] = 0.5 * (b.lo[a] + b.hi[a]) - o[a];
        }
        if (r % 7 == 0) d[r % 3] = 0.0;  // axis-parallel components
        std::vector<int> reached(n_tree_items, 0);
        std::vector<uint32_t> todo{0u};
        while (!todo.empty()) {
            const uint32_t ref = todo.back();
            todo.pop_back();
            if (ref & rtxd::kBvhLeaf) {
                const uint32_t first = ref & rtxd::kBvhLeafFirst, count = (ref >> rtxd::kBvhLeafCountShift) & 15u;
                for (uint32_t k = 0; k < count; ++k) reached[first + k] = 1;
                continue;
            }
            const rtxd::BvhNode& N = tree.nodes[ref];
            double lo[3], hi[3];
            const auto planes = [](const uint16_t q[3][2], int k, uint16_t out[3]) { for (int a = 0; a < 3; ++a) out[a] = q[a][k]; };
            uint16_t ql[3], qh[3];
            planes(N.box0, 0, ql); planes(N.box0, 1, qh);
            decode(ql, lo); decode(qh, hi);
            if (ray_box(o, d, lo, hi)) todo.push_back(N.child0);
            planes(N.box1, 0, ql); planes(N.box1, 1, qh);
            decode(ql, lo); decode(qh, hi);
            if (ray_box(o, d, lo, hi)) todo.push_back(N.child1);
        }
        for (uint32_t k = 0; k < n_tree_items; ++k) {
            const Box3& b = exact[item_group[k]];
            if (ray_box(o, d, b.lo, b.hi)) {
                ++checked;
                if (!reached[k]) return fail("a ray meets an item's bounds but the walk does not reach its leaf");
            }
        }
    }

    // the builder's surface-area estimate of the inner nodes a ray through the root box visits: the
    // root counts once
    if (tree.nodes.size() > 1 && !(tree.visits >= 1.0 - 1e-9)) return fail("visit estimate: expected >= 1");
    std::printf("estimated visits=%.2f\n", tree.visits);

    // ---- light tree: every group as a light leaf, entries = arbitrary distinct numbers ----
    std::vector<uint32_t> entries(groups.size());
    for (size_t g = 0; g < groups.size(); ++g) entries[g] = 1000u + 3u * (uint32_t)g;
    std::vector<rtxd::LightBvhNode> lt;
    rtxbvh::build_light_tree(prims, groups, steps, entries, lt);
    if (lt.size() != 2 * groups.size() - 1) return fail("light tree: node count != 2 n - 1");
    std::vector<int> lseen(groups.size(), 0);
    for (size_t i = 0; i < lt.size(); ++i) {
        const rtxd::LightBvhNode& N = lt[i];
        if (N.skip <= i || N.skip > lt.size()) return fail("light tree: skip link does not point forward");
        if (N.entry != rtxd::kLightBvhInner) {
            if (N.skip != i + 1) return fail("light tree: a leaf's skip link is not the next node");
            if (N.entry < 1000u || (N.entry - 1000u) % 3u || (N.entry - 1000u) / 3u >= groups.size()) return fail("light tree: bad entry");
            ++lseen[(N.entry - 1000u) / 3u];
        } else {
            // children: i + 1 and lt[i + 1].skip; both inside the parent's box and subtree
            const size_t c0 = i + 1, c1 = lt[c0].skip;
            if (c1 >= N.skip || lt[c1].skip != N.skip) return fail("light tree: children do not tile the subtree");
            for (size_t c : {c0, c1})
                for (int a = 0; a < 3; ++a)
                    if (lt[c].lo[a] < N.lo[a] || lt[c].hi[a] > N.hi[a]) return fail("light tree: child box outside its parent's");
        }
    }
    for (int s : lseen)
        if (s != 1) return fail("light tree: an entry is not in exactly one leaf");
    uint32_t lchecked = 0, visited = 0;
    for (int r = 0; r < 400; ++r) {
        const double o[3] = {100.0 * U(gen), 10.0 * U(gen) + 5.0, 100.0 * U(gen)};
        double d[3] = {D(gen), D(gen), D(gen)};
        if (r % 2 == 0) {
            const Box3& b = exact[gen() % groups.size()];
            for (int a = 0; a < 3; ++a) d[a] = 0.5 * (b.lo[a] + b.hi[a]) - o[a];
        }
        if (r % 5 == 0) d[r % 3] = 0.0;
        std::vector<int> reached(groups.size(), 0);
        for (uint32_t i = 0; i < lt.size();) {  // the device's walk
            const rtxd::LightBvhNode& N = lt[i];
            const double lo[3] = {N.lo[0], N.lo[1], N.lo[2]}, hi[3] = {N.hi[0], N.hi[1], N.hi[2]};
            ++visited;
            if (!ray_box(o, d, lo, hi)) { i = N.skip; continue; }
            if (N.entry != rtxd::kLightBvhInner) reached[(N.entry - 1000u) / 3u] = 1;
            ++i;
        }
        for (size_t g = 0; g < groups.size(); ++g)
            if (ray_box(o, d, exact[g].lo, exact[g].hi)) {
                ++lchecked;
                if (!reached[g]) return fail("light tree: a ray meets a light's bounds but the walk skips it");
            }
    }
    std::printf("ok items=%zu nodes=%zu big=%u depth=%d ray-item pairs=%u | light nodes=%zu pairs=%u visited/ray=%.1f\n",
                tree.items.size(), tree.nodes.size(), tree.n_big, max_depth, checked, lt.size(), lchecked, visited / 400.0);
    return 0;
}
