// rtx_bvh.h — host-side BVH builder over scene groups (a bare primitive or one
// transformed instance such as a Box under Rotate / Translate).
//
// The reference tracer has no acceleration structure: ShapeList::hit
// (Raytrace_C++/Src/ShapeList.cpp:5-17) scans every primitive.  The tree built
// here only prunes that scan; the traversal kernel resolves ties between equal
// hit distances by the list-order rule of the scan, so it returns the same
// nearest hit.
#pragma once
#include <string>
#include <vector>

#include "rtx.h"
#include "rtx_device.h"

namespace rtxbvh {

// Maximum tree depth the traversal stack supports (see kBvhStack in rtx_core.h).
constexpr int kMaxDepth = 21;
static_assert(kMaxDepth + 1 <= rtxd::kBvhStackEntries, "the traversal stack must hold one deferred child per level");

// What the builder hands back.
//   nodes   two-child nodes with 16-bit boxes on the grid `origin + q * scale` (rtxd::BvhNode),
//           nodes[0] is the root; every box is rounded outwards by at least one grid step
//   items   the groups in leaf order, followed by `n_big` groups that are NOT in the tree:
//           items far larger than the typical one (a ground sphere of radius 1000 under spheres
//           of radius 0.2, a floor under 100 000 small objects).  Every ray tests them directly;
//           inside the tree they would overlap everything and, worse, stretch the grid the
//           boxes are quantised on.
struct Tree {
    std::vector<rtxd::BvhNode> nodes;
    std::vector<rtxd::Group> items;
    uint32_t n_big = 0;
    int depth = 0;  // inner-node levels above the deepest leaf
    // surface-area estimate: the inner nodes a ray through the root box visits on average (sum of
    // the inner nodes' areas over the root's area)
    double visits = 0.0;
    float origin[3] = {0, 0, 0};
    float scale[3] = {1, 1, 1};
};

// The caller's flattened scene as device records: primitives, transform steps, and the groups -
// runs of consecutive primitives under one transform chain - that the linear scan walks and the
// tree is built over.  For the scan (`for_tree` false) a run of untransformed primitives is one
// group; for the tree every untransformed primitive is an item of its own, so that the tree can
// separate them, and a transformed run (a Box under Rotate / Translate) stays one item.  Returns
// false (and sets err) when a transformed run is too long for a leaf record's count.
bool scene_groups(const RtxScene* s, bool for_tree, std::vector<rtxd::Prim>& prims, std::vector<rtxd::Step>& steps,
                  std::vector<rtxd::Group>& groups, std::string& err);

// Returns false (and sets err) if the tree cannot be built within kMaxDepth.
bool build(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::Group>& groups,
           const std::vector<rtxd::Step>& steps, Tree& out, std::string& err);

// Bounds of the whole scene (every group's padded world-space box, as the trees use them), rounded
// outwards to fp32: a ray that does not meet them meets no primitive.
void scene_bounds(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::Group>& groups,
                  const std::vector<rtxd::Step>& steps, float lo[3], float hi[3]);

// Tree over the light leaves for the density sum (rtxd::LightBvhNode): group g of `groups` is
// one light leaf (its shape as a primitive, its transform chain), `entries[g]` the index the
// leaf's node carries.  Median splits on the widest axis of the centres, one leaf per node,
// boxes padded like the scene tree's; nodes come out in depth-first order with skip links.
void build_light_tree(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::Group>& groups,
                      const std::vector<rtxd::Step>& steps, const std::vector<uint32_t>& entries,
                      std::vector<rtxd::LightBvhNode>& out);

}  // namespace rtxbvh
