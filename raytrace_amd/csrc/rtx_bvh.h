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

#include "rtx_device.h"

namespace rtxbvh {

// Maximum tree depth the traversal stack supports (see kBvhStack in rtx_core.h).
constexpr int kMaxDepth = 22;

// nodes[0] is the root (two-child nodes, see rtxd::BvhNode); `items` are the groups in leaf
// order.  Returns false (and sets err) if the tree cannot be built within kMaxDepth.
bool build(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::Group>& groups,
           const std::vector<rtxd::Step>& steps, std::vector<rtxd::BvhNode>& nodes,
           std::vector<rtxd::Group>& items, std::string& err);

}  // namespace rtxbvh
