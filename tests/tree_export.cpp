// tree_export.cpp — hands the scene tree of the renderer's own builder (raytrace_amd/csrc/rtx_bvh.cpp:
// scene_groups + build, the calls rtx_upload_scene makes) to Python as plain arrays, with the 16-bit
// boxes decoded to world coordinates.  Test infrastructure (host only, compiled with g++ by
// tests/golden/make_golden.py): the op-counting run of the CPU restatement walks this tree, so that the
// flops a ray is priced with are those of the tree the GPU kernel walks.
//
// Node layout handed out: one record per box.  Record 0 is the root (its box is the union of its
// children's and is never tested); an inner record has left / right >= 0, a leaf record has
// count == 1 and first = the index of its item; items are (prim_begin, prim_end) ranges; the last
// n_big items are the ones the builder keeps outside the tree (every ray tests them first).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "rtx.h"
#include "rtx_bvh.h"

namespace {
struct Out {
    std::vector<float> lo, hi;  // 3 per record
    std::vector<int32_t> left, right, first, count;
    std::vector<uint32_t> item_begin, item_end;
    uint32_t n_big = 0;
    int depth = 0;
    double visits = 0.0;
} g;

void decode(const uint16_t q[3][2], const rtxbvh::Tree& t, float lo[3], float hi[3]) {
    for (int a = 0; a < 3; ++a) {
        lo[a] = t.origin[a] + (float)q[a][0] * t.scale[a];
        hi[a] = t.origin[a] + (float)q[a][1] * t.scale[a];
    }
}

int add_record(const float lo[3], const float hi[3]) {
    for (int a = 0; a < 3; ++a) {
        g.lo.push_back(lo[a]);
        g.hi.push_back(hi[a]);
    }
    g.left.push_back(-1);
    g.right.push_back(-1);
    g.first.push_back(0);
    g.count.push_back(0);
    return (int)g.left.size() - 1;
}
}  // namespace

extern "C" {

// Builds the tree of `scene`; returns the number of records (0 on failure), and the item counts.
int tree_export_build(const RtxScene* scene, uint32_t* n_items, uint32_t* n_big, int32_t* depth, double* visits) {
    g = Out{};
    std::vector<rtxd::Prim> prims;
    std::vector<rtxd::Step> steps;
    std::vector<rtxd::Group> groups;
    std::string err;
    if (!rtxbvh::scene_groups(scene, true, prims, steps, groups, err)) return 0;
    rtxbvh::Tree tree;
    if (!rtxbvh::build(prims, groups, steps, tree, err)) return 0;
    for (const rtxd::Group& G : tree.items) {
        g.item_begin.push_back(G.prim_begin);
        g.item_end.push_back(G.prim_end);
    }
    g.n_big = tree.n_big;
    g.depth = tree.depth;
    g.visits = tree.visits;
    // record of product node k: made when its parent is converted; the root first
    std::vector<int> record_of(tree.nodes.size(), -1);
    const float zero[3] = {0, 0, 0};
    record_of[0] = add_record(zero, zero);
    std::vector<uint32_t> todo{0u};
    while (!todo.empty()) {
        const uint32_t k = todo.back();
        todo.pop_back();
        const rtxd::BvhNode& N = tree.nodes[k];
        const uint32_t child[2] = {N.child0, N.child1};
        int rec[2];
        for (int c = 0; c < 2; ++c) {
            float lo[3], hi[3];
            decode(c == 0 ? N.box0 : N.box1, tree, lo, hi);
            rec[c] = add_record(lo, hi);
            if (child[c] & rtxd::kBvhLeaf) {
                g.first[rec[c]] = (int32_t)(child[c] & rtxd::kBvhLeafFirst);
                g.count[rec[c]] = (int32_t)((child[c] >> rtxd::kBvhLeafCountShift) & 15u);
            } else {
                record_of[child[c]] = rec[c];
                todo.push_back(child[c]);
            }
        }
        g.left[record_of[k]] = rec[0];
        g.right[record_of[k]] = rec[1];
    }
    *n_items = (uint32_t)g.item_begin.size();
    *n_big = g.n_big;
    *depth = g.depth;
    *visits = g.visits;
    return (int)g.left.size();
}

void tree_export_copy(float* lo, float* hi, int32_t* left, int32_t* right, int32_t* first, int32_t* count,
                      uint32_t* item_begin, uint32_t* item_end) {
    std::memcpy(lo, g.lo.data(), g.lo.size() * sizeof(float));
    std::memcpy(hi, g.hi.data(), g.hi.size() * sizeof(float));
    std::memcpy(left, g.left.data(), g.left.size() * sizeof(int32_t));
    std::memcpy(right, g.right.data(), g.right.size() * sizeof(int32_t));
    std::memcpy(first, g.first.data(), g.first.size() * sizeof(int32_t));
    std::memcpy(count, g.count.data(), g.count.size() * sizeof(int32_t));
    std::memcpy(item_begin, g.item_begin.data(), g.item_begin.size() * sizeof(uint32_t));
    std::memcpy(item_end, g.item_end.data(), g.item_end.size() * sizeof(uint32_t));
}
}
