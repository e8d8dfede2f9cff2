// rtx_bvh.cpp — binned-SAH top-down BVH build over scene groups (host, C++17).
#include "rtx_bvh.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "rtx.h"

namespace rtxbvh {
namespace {

struct Box {
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX};
    double hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    void add(const double p[3]) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void add(const Box& b) {
        add(b.lo);
        add(b.hi);
    }
    double area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

void quat_rotate(const rtxd::Step& q, double v[3]) {
    // v' = q v q*, expanded in double precision (bounds only; not on the hit path)
    double x = q.x, y = q.y, z = q.z, w = q.w;
    double tx = w * v[0] + y * v[2] - z * v[1];
    double ty = w * v[1] + z * v[0] - x * v[2];
    double tz = w * v[2] + x * v[1] - y * v[0];
    double tw = x * v[0] + y * v[1] + z * v[2];
    double rx = tw * x + tx * w - ty * z + tz * y;
    double ry = tw * y + ty * w - tz * x + tx * z;
    double rz = tw * z + tz * w - tx * y + ty * x;
    v[0] = rx;
    v[1] = ry;
    v[2] = rz;
}

Box prim_box(const rtxd::Prim& P) {
    Box b;
    const uint32_t kind = P.kind_flip & 3u;
    if (kind == RTX_PRIM_SPHERE) {
        double r = std::fabs((double)P.d);
        double lo[3] = {P.a - r, P.b - r, P.c - r}, hi[3] = {P.a + r, P.b + r, P.c + r};
        b.add(lo);
        b.add(hi);
        return b;
    }
    int xi, yi, zi;
    if (kind == RTX_PRIM_RECT_XY) { xi = 0; yi = 1; zi = 2; }
    else if (kind == RTX_PRIM_RECT_XZ) { xi = 0; yi = 2; zi = 1; }
    else { xi = 1; yi = 2; zi = 0; }
    double lo[3], hi[3];
    lo[xi] = std::min(P.a, P.b); hi[xi] = std::max(P.a, P.b);
    lo[yi] = std::min(P.c, P.d); hi[yi] = std::max(P.c, P.d);
    lo[zi] = hi[zi] = P.e;
    b.add(lo);
    b.add(hi);
    return b;
}

Box group_box(const std::vector<rtxd::Prim>& prims, const rtxd::Group& G,
              const std::vector<rtxd::Step>& steps) {
    Box local;
    for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) local.add(prim_box(prims[p]));
    Box world;
    for (int c = 0; c < 8; ++c) {
        double v[3] = {(c & 1) ? local.hi[0] : local.lo[0], (c & 2) ? local.hi[1] : local.lo[1],
                       (c & 4) ? local.hi[2] : local.lo[2]};
        for (uint32_t s = G.step_end; s > G.step_begin; --s) {  // innermost wrapper first
            const rtxd::Step& S = steps[s - 1];
            if (S.kind == RTX_STEP_ROTATE) quat_rotate(S, v);
            else { v[0] += S.x; v[1] += S.y; v[2] += S.z; }
        }
        world.add(v);
    }
    // Pad: the traversal's slab test runs in fp32 with approximate reciprocals.
    for (int a = 0; a < 3; ++a) {
        double mag = std::max({std::fabs(world.lo[a]), std::fabs(world.hi[a]), world.hi[a] - world.lo[a]});
        double pad = 4e-4 * mag + 1e-5;
        world.lo[a] -= pad;
        world.hi[a] += pad;
    }
    return world;
}

struct BuildItem {
    Box box;
    double centre[3];
    uint32_t group;
};

struct Task {
    uint32_t node, begin, end;
    int depth;
};

// Intermediate one-box-per-node tree: inner nodes keep their children adjacent.
struct TmpNode {
    Box box;
    uint32_t left = 0, count = 0;  // count == 0: inner, children left / left+1; else leaf items [left, left+count)
};

// The grid the boxes are stored on: origin + q * scale, q = 0 .. 65535 per axis.
struct Grid {
    float origin[3], scale[3];
};

Grid make_grid(const Box& root) {
    Grid g;
    for (int a = 0; a < 3; ++a) {
        const double ext = std::max(root.hi[a] - root.lo[a], 1e-30);
        // two steps of slack at either end for the outward rounding below
        g.scale[a] = std::nextafterf((float)(ext / 65530.0), FLT_MAX);
        g.origin[a] = std::nextafterf((float)(root.lo[a] - 2.0 * g.scale[a]), -FLT_MAX);
    }
    return g;
}

// Outward rounding by one whole step beyond floor / ceil: the traversal evaluates
// origin + q * scale folded into the ray's slab constants in fp32, whose rounding error is
// orders of magnitude below a grid step (1 / 65530 of the scene's extent).
void store_box(uint16_t q[3][2], const Box& b, const Grid& g) {
    for (int a = 0; a < 3; ++a) {
        const double l = std::floor((b.lo[a] - (double)g.origin[a]) / (double)g.scale[a]) - 1.0;
        const double h = std::ceil((b.hi[a] - (double)g.origin[a]) / (double)g.scale[a]) + 1.0;
        q[a][0] = (uint16_t)std::min(std::max(l, 0.0), 65535.0);
        q[a][1] = (uint16_t)std::min(std::max(h, 0.0), 65535.0);
    }
}

}  // namespace

bool scene_groups(const RtxScene* s, bool for_tree, std::vector<rtxd::Prim>& prims, std::vector<rtxd::Step>& steps,
                  std::vector<rtxd::Group>& groups, std::string& err) {
    prims.assign(s->n_prims, rtxd::Prim{});
    for (uint32_t p = 0; p < s->n_prims; ++p) {
        rtxd::Prim& P = prims[p];
        P.a = s->prim_a[p]; P.b = s->prim_b[p]; P.c = s->prim_c[p]; P.d = s->prim_d[p]; P.e = s->prim_e[p];
        P.kind_flip = s->prim_kind[p] | (s->prim_flip[p] ? 4u : 0u);
        P.material = s->prim_material[p];
        P.group = 0;
    }
    steps.assign(s->n_steps, rtxd::Step{});
    for (uint32_t k = 0; k < s->n_steps; ++k) {
        rtxd::Step& S = steps[k];
        S.x = s->step_x[k]; S.y = s->step_y[k]; S.z = s->step_z[k]; S.w = s->step_w[k];
        S.kind = s->step_kind[k];
        S.pad[0] = S.pad[1] = S.pad[2] = 0;
    }
    groups.clear();
    for (uint32_t p = 0; p < s->n_prims;) {
        const uint32_t c = s->prim_chain[p];
        uint32_t q = p + 1;
        if (!(for_tree && c == 0))
            while (q < s->n_prims && s->prim_chain[q] == c) ++q;
        for (uint32_t k = p; k < q; ++k) prims[k].group = (uint32_t)groups.size();
        if (for_tree && q - p >= (1u << 24)) {  // a leaf record counts the primitives of a group in 24 bits
            err = "more than 2^24 primitives under one transform chain";
            return false;
        }
        groups.push_back(rtxd::Group{p, q, s->chain_begin[c], s->chain_begin[c + 1]});
        p = q;
    }
    return true;
}

bool build(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::Group>& groups,
           const std::vector<rtxd::Step>& steps, Tree& out, std::string& err) {
    std::vector<TmpNode> nodes;
    std::vector<rtxd::BvhNode>& out_nodes = out.nodes;
    std::vector<rtxd::Group>& items = out.items;
    out_nodes.clear();
    const uint32_t n_all = (uint32_t)groups.size();
    items.clear();
    out.n_big = 0;
    out.depth = 0;
    if (n_all == 0) {
        err = "no groups";
        return false;
    }
    std::vector<BuildItem> it(n_all);
    std::vector<double> extent(n_all);
    for (uint32_t g = 0; g < n_all; ++g) {
        it[g].box = group_box(prims, groups[g], steps);
        it[g].group = g;
        for (int a = 0; a < 3; ++a) it[g].centre[a] = 0.5 * (it[g].box.lo[a] + it[g].box.hi[a]);
        extent[g] = std::max({it[g].box.hi[0] - it[g].box.lo[0], it[g].box.hi[1] - it[g].box.lo[1],
                              it[g].box.hi[2] - it[g].box.lo[2]});
    }
    // Items far larger than the typical one stay out of the tree (at most kMaxBig of them, the
    // largest first): see Tree.
    constexpr uint32_t kMaxBig = 8;
    std::vector<uint32_t> big;
    if (n_all >= 32) {
        std::vector<double> sorted = extent;
        std::nth_element(sorted.begin(), sorted.begin() + n_all / 2, sorted.end());
        const double limit = 32.0 * sorted[n_all / 2];
        for (uint32_t g = 0; g < n_all; ++g)
            if (extent[g] > limit) big.push_back(g);
        std::sort(big.begin(), big.end(), [&](uint32_t x, uint32_t y) { return extent[x] > extent[y]; });
        if (big.size() > kMaxBig) big.resize(kMaxBig);
        std::sort(big.begin(), big.end());
        for (size_t k = big.size(); k-- > 0;) it.erase(it.begin() + big[k]);
    }
    const uint32_t n = (uint32_t)it.size();
    // one item per leaf: the leaf code (four record types, each its own branch) is the expensive
    // part of a traversal step on the GPU; measured 100k-item scene 4.67 (2 items) -> 5.22 Grays/s,
    // 487-sphere field 21.1 -> 20.6
    constexpr uint32_t kLeafItems = 1;
    constexpr int kBins = 16;
    nodes.reserve(2 * (size_t)n);
    nodes.emplace_back();
    std::vector<Task> todo;
    todo.push_back(Task{0, 0, n, 0});
    while (!todo.empty()) {
        Task t = todo.back();
        todo.pop_back();
        Box bounds, cbounds;
        for (uint32_t k = t.begin; k < t.end; ++k) {
            bounds.add(it[k].box);
            cbounds.add(it[k].centre);
        }
        const uint32_t count = t.end - t.begin;
        if (count <= kLeafItems) {
            nodes[t.node].box = bounds;
            nodes[t.node].left = t.begin;
            nodes[t.node].count = count;
            out.depth = std::max(out.depth, t.depth);
            continue;
        }
        if (t.depth >= kMaxDepth) {
            err = "BVH deeper than the traversal stack allows";
            return false;
        }
        // binned SAH over the widest centroid axis first, then the others
        int best_axis = -1, best_bin = -1;
        double best_cost = DBL_MAX;
        // Once the remaining depth budget only just covers a balanced split, stop
        // trusting SAH and split at the median.
        const bool force_median = (1u << std::min(kMaxDepth - t.depth - 1, 31)) < count;
        if (!force_median) {
            for (int a = 0; a < 3; ++a) {
                double lo = cbounds.lo[a], ext = cbounds.hi[a] - cbounds.lo[a];
                if (!(ext > 0.0)) continue;
                Box bb[kBins];
                uint32_t bc[kBins] = {0};
                for (uint32_t k = t.begin; k < t.end; ++k) {
                    int b = (int)((it[k].centre[a] - lo) / ext * kBins);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bb[b].add(it[k].box);
                    bc[b]++;
                }
                double right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    if (bc[b]) acc.add(bb[b]);
                    cnt += bc[b];
                    right_area[b] = acc.area();
                    right_cnt[b] = cnt;
                }
                Box lacc;
                uint32_t lcnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    if (bc[b]) lacc.add(bb[b]);
                    lcnt += bc[b];
                    if (lcnt == 0 || right_cnt[b + 1] == 0) continue;
                    double cost = lacc.area() * lcnt + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_bin = b;
                    }
                }
            }
        }
        uint32_t mid;
        if (best_axis >= 0) {
            const int a = best_axis;
            const double lo = cbounds.lo[a], ext = cbounds.hi[a] - cbounds.lo[a];
            auto first_right = std::partition(it.begin() + t.begin, it.begin() + t.end, [&](const BuildItem& x) {
                int b = (int)((x.centre[a] - lo) / ext * kBins);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= best_bin;
            });
            mid = (uint32_t)(first_right - it.begin());
        } else {
            int a = 0;
            for (int k = 1; k < 3; ++k)
                if (cbounds.hi[k] - cbounds.lo[k] > cbounds.hi[a] - cbounds.lo[a]) a = k;
            mid = t.begin + count / 2;
            std::nth_element(it.begin() + t.begin, it.begin() + mid, it.begin() + t.end,
                             [a](const BuildItem& x, const BuildItem& y) { return x.centre[a] < y.centre[a]; });
        }
        if (mid == t.begin || mid == t.end) mid = t.begin + count / 2;
        const uint32_t left = (uint32_t)nodes.size();
        nodes.emplace_back();
        nodes.emplace_back();
        nodes[t.node].box = bounds;
        nodes[t.node].left = left;
        nodes[t.node].count = 0;
        todo.push_back(Task{left, t.begin, mid, t.depth + 1});
        todo.push_back(Task{left + 1, mid, t.end, t.depth + 1});
    }
    items.resize(n);
    for (uint32_t k = 0; k < n; ++k) items[k] = groups[it[k].group];
    for (uint32_t g : big) items.push_back(groups[g]);
    out.n_big = (uint32_t)big.size();
    const Grid grid = make_grid(nodes[0].box);
    for (int a = 0; a < 3; ++a) {
        out.origin[a] = grid.origin[a];
        out.scale[a] = grid.scale[a];
    }

    // Convert to two-child nodes.  A child reference is a node index or a tagged leaf.
    std::vector<uint32_t> inner_index(nodes.size(), 0);
    uint32_t n_inner = 0;
    for (size_t k = 0; k < nodes.size(); ++k)
        if (nodes[k].count == 0) inner_index[k] = n_inner++;
    auto ref_of = [&](uint32_t k) -> uint32_t {
        const TmpNode& N = nodes[k];
        if (N.count == 0) return inner_index[k];
        return rtxd::kBvhLeaf | (N.count << rtxd::kBvhLeafCountShift) | N.left;
    };
    if (n > rtxd::kBvhLeafFirst) {
        err = "too many BVH items";
        return false;
    }
    static_assert(kLeafItems == 1, "the fast trace kernel reads one record per leaf");
    {   // expected node visits of a ray through the root box, by the surface-area measure
        const double root = nodes[0].box.area();
        double sum = 0.0;
        for (const TmpNode& N : nodes)
            if (N.count == 0) sum += N.box.area();
        out.visits = root > 0.0 ? sum / root : 0.0;
    }
    if (nodes[0].count != 0) {
        // a single leaf: root node with one real child and one empty box
        // (the leaf is then visited twice when the ray meets its box: harmless, a primitive
        // that already holds the nearest hit does not win against itself)
        rtxd::BvhNode R{};
        store_box(R.box0, nodes[0].box, grid);
        store_box(R.box1, nodes[0].box, grid);
        R.child0 = ref_of(0);
        R.child1 = R.child0;
        out_nodes.push_back(R);
        return true;
    }
    out_nodes.resize(n_inner);
    for (size_t k = 0; k < nodes.size(); ++k) {
        const TmpNode& N = nodes[k];
        if (N.count != 0) continue;
        rtxd::BvhNode& O = out_nodes[inner_index[k]];
        O = rtxd::BvhNode{};
        store_box(O.box0, nodes[N.left].box, grid);
        store_box(O.box1, nodes[N.left + 1].box, grid);
        O.child0 = ref_of(N.left);
        O.child1 = ref_of(N.left + 1);
    }
    return true;
}

void scene_bounds(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::Group>& groups,
                  const std::vector<rtxd::Step>& steps, float lo[3], float hi[3]) {
    Box all;
    for (const rtxd::Group& G : groups) all.add(group_box(prims, G, steps));
    for (int a = 0; a < 3; ++a) {
        lo[a] = std::nextafterf((float)all.lo[a], -FLT_MAX);
        hi[a] = std::nextafterf((float)all.hi[a], FLT_MAX);
    }
}

namespace {
struct LightItem {
    Box box;
    double centre[3];
    uint32_t entry;
};

// Emits the subtree over items [begin, end) at out.size() and returns nothing: the caller reads
// the subtree's extent from out.size() afterwards.
void emit_light_subtree(std::vector<LightItem>& items, size_t begin, size_t end, std::vector<rtxd::LightBvhNode>& out) {
    Box box;
    for (size_t k = begin; k < end; ++k) box.add(items[k].box);
    const size_t self = out.size();
    out.emplace_back();
    for (int a = 0; a < 3; ++a) {
        out[self].lo[a] = std::nextafterf((float)box.lo[a], -FLT_MAX);  // double -> float, outwards
        out[self].hi[a] = std::nextafterf((float)box.hi[a], FLT_MAX);
    }
    if (end - begin == 1) {
        out[self].entry = items[begin].entry;
    } else {
        out[self].entry = rtxd::kLightBvhInner;
        Box centres;
        for (size_t k = begin; k < end; ++k) centres.add(items[k].centre);
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (centres.hi[a] - centres.lo[a] > centres.hi[axis] - centres.lo[axis]) axis = a;
        const size_t mid = begin + (end - begin) / 2;
        std::nth_element(items.begin() + begin, items.begin() + mid, items.begin() + end,
                         [axis](const LightItem& x, const LightItem& y) { return x.centre[axis] < y.centre[axis]; });
        emit_light_subtree(items, begin, mid, out);  // depth log2(n): the halves are equal
        emit_light_subtree(items, mid, end, out);
    }
    out[self].skip = (uint32_t)out.size();
}
}  // namespace

void build_light_tree(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::Group>& groups,
                      const std::vector<rtxd::Step>& steps, const std::vector<uint32_t>& entries,
                      std::vector<rtxd::LightBvhNode>& out) {
    out.clear();
    std::vector<LightItem> items(groups.size());
    for (size_t g = 0; g < groups.size(); ++g) {
        items[g].box = group_box(prims, groups[g], steps);
        for (int a = 0; a < 3; ++a) items[g].centre[a] = 0.5 * (items[g].box.lo[a] + items[g].box.hi[a]);
        items[g].entry = entries[g];
    }
    if (!items.empty()) emit_light_subtree(items, 0, items.size(), out);
}

}  // namespace rtxbvh
