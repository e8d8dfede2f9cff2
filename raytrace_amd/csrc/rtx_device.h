// rtx_device.h — the scene as it lies in HBM and the kernel argument blocks.
//
// The C ABI hands the scene over as struct-of-arrays (include/rtx.h); at upload the
// library repacks it into 32-byte records.  Every lane of a wave walks the SAME
// primitive / group / light list, so record fields are fetched with wave-uniform
// scalar loads (one s_load_dwordx8 per record) and used as SGPR operands; only the
// per-lane winner of a nearest-hit query is re-read with a vector load.
#pragma once
#include <stdint.h>

namespace rtxd {

// One world primitive.  Rect (Rect.h:30): a=x0 b=x1 c=y0 d=y1 e=k.  Sphere
// (Sphere.h:19-20): a,b,c = centre, d = radius.
struct alignas(32) Prim {
    float a, b, c, d, e;
    uint32_t kind_flip;  // bits 0-1: RTX_PRIM_*; bit 2: FlipNormals parity
    uint32_t material;   // index into Material[]
    uint32_t group;      // index into Group[] (the transform chain this primitive sits under)
};

// Everything the integrator needs about the primitive that won a nearest-hit query, in one
// record, so that completing the hit and shading it costs one per-lane load instead of a
// chain of dependent ones (primitive -> group -> steps, primitive -> material -> texture).
struct alignas(32) PrimInfo {
    float r, g, b;        // albedo / emission when the material's texture is a plain colour
    float param;          // Metal fuzz | Dielectric refractive index
    uint32_t bits;        // see kInfo* below
    uint32_t texture;     // texture index when kInfoTextured is set
    uint32_t step_begin;  // transform chain of the primitive's group, outermost wrapper first
    uint32_t step_end;
};
constexpr uint32_t kInfoKindMask = 3u;       // RTX_PRIM_*
constexpr uint32_t kInfoFlip = 4u;           // FlipNormals parity
constexpr uint32_t kInfoMatShift = 3u;       // RTX_MAT_* in bits 3-4
constexpr uint32_t kInfoTextured = 32u;      // texture is not a plain colour: evaluate it
constexpr uint32_t kInfoNeedsUV = 64u;       // ... and it (or a sub-texture) is an image: needs (u, v)

// Fast-mode linear scan: the primitives of a group re-ordered by kind (XY rects, XZ rects,
// YZ rects, spheres) so that each kind is scanned by a branch-free loop over homogeneous
// 32-byte records fetched with one scalar load each.
struct alignas(32) ScanRec {
    float a, b, c, d, e, f;  // rect: x0 x1 y0 y1 k | sphere: cx cy cz r r*r | box: lo xyz, hi xyz
    uint32_t pad[2];
};
// One group of the bucketed table: the composed world->group affine map (rows of the 3x3
// matrix then the offset) instead of the un-composed step chain, and the kind counts.
// Six rects of a group that are the complete faces of one axis-aligned box (what Box.cpp:11-29
// builds) are scanned as ONE record by a slab test.  A box owns six consecutive positions of
// the table - the record itself, then five unused ones - so that position + face (0..5 = +z,
// -z, +y, -y, +x, -x, the order of Box.cpp:15-28) still names one ScanHit / original primitive.
struct alignas(16) ScanGroup {
    float m[12];           // p_local = (m[0..2].p + m[9], m[3..5].p + m[10], m[6..8].p + m[11])
    uint32_t first;        // index of the group's first ScanRec
    uint32_t n_xy, n_xz, n_yz, n_sph;
    uint32_t identity;     // 1: no transform
    uint32_t n_box;
    uint32_t rot_y;        // 1: the map is a rotation about the y axis and a translation (m[1] = m[3] =
                           // m[5] = m[7] = 0 and m[4] = 1 exactly): the terms that vanish are not computed
};
constexpr uint32_t kBoxSlots = 6;

// What completing a fast-mode scan hit needs, in world space, so that no transform is applied
// per hit: n = (p - c) * inv_r + n0 covers both kinds (rect: inv_r = 0, n0 = the rect's
// normal taken through the group's rotation, FlipNormals parity applied; sphere: c = centre
// in world space, inv_r = +-1 / radius, n0 = 0).  Rotations and translations keep lengths,
// so (p - c) / r is the same vector whether it is formed in the group's frame and rotated
// out (Sphere.cpp:19, Rotate.cpp:13) or formed in world space.
struct alignas(64) ScanHit {
    float r, g, b, param;  // as PrimInfo
    uint32_t bits, texture;
    uint32_t group;        // scan group (its affine map is only needed for texture coordinates)
    uint32_t chained;      // 1: the group has a transform
    float n0[3], inv_r;
    float c[3], pad;
};

// A run of consecutive primitives that share one transform chain.
struct alignas(16) Group {
    uint32_t prim_begin, prim_end;
    uint32_t step_begin, step_end;  // outermost wrapper first
};

// One Translate (x,y,z = offset) or Rotate (x,y,z,w = unit quaternion) wrapper.
struct alignas(32) Step {
    float x, y, z, w;
    uint32_t kind;  // RTX_STEP_*
    uint32_t pad[3];
};

struct alignas(16) Material {
    uint32_t kind;     // RTX_MAT_*
    uint32_t texture;  // albedo / emission
    float param;       // fuzz | refractive index
    uint32_t pad;
};

struct alignas(32) Texture {
    uint32_t kind;       // RTX_TEX_*
    float r, g, b;       // colour; checker: r = frequency
    uint32_t odd, even;  // checker: sub-textures; image: width, height
    uint32_t offset;     // image: byte offset of the first texel
    uint32_t pad;
};

struct alignas(32) Light {
    float a, b, c, d, e;  // as Prim
    uint32_t kind;        // RTX_LIGHT_*
    float inv_area;       // rect: 1 / ((x1 - x0) * (y1 - y0)), used by the fast-mode pdf
    uint32_t pad;
};

// Light trees and general light shapes (kernels instantiated with kLightTree only): the entry's
// place in the tree and its transform chain, index-aligned with Light[].
struct alignas(32) LightNode {
    uint32_t first, count;          // RTX_LIGHT_LIST: children are entries [first, first + count)
    uint32_t step_begin, step_end;  // leaf: transform chain, outermost wrapper first
    float weight;                   // product of the 1 / n of every list above the entry
    uint32_t pad[3];
};
// Many lights: a bounding-volume tree over the light leaves in depth-first order with skip
// links.  ShapeList::pdf_value (ShapeList.cpp:19-26) asks EVERY entry for its density along
// (o, v), and an entry answers 0 unless the ray meets it (Rect.cpp:38-40, Sphere.cpp:40-42): the
// sum only has to visit the leaves whose box the ray meets.  A query wants all of them, in any
// order, so it needs no stack: at node i go to i + 1 (the first child, or past a leaf) when the
// ray meets the box, to `skip` (the node after i's whole subtree) when it does not.
struct alignas(32) LightBvhNode {
    float lo[3];
    uint32_t skip;   // index of the node after this one's subtree (n_light_bvh = the end)
    float hi[3];
    uint32_t entry;  // leaf: index into lights[] / light_nodes[]; inner node: kLightBvhInner
};
constexpr uint32_t kLightBvhInner = 0xffffffffu;
#ifndef RTX_LIGHT_BVH_MIN
// Light leaves from which the tree is built and used.  Below it the plain sum over wave-uniform
// (scalar-loaded) records is faster than the per-lane walk: measured on a Cornell room with k x k
// ceiling lamps, 68 lights 23.8 ms (sum) / 25.8 ms (tree), 104 lights 30.4 / 28.8, 260 lights
// 46.9 / 27.9, 1 604 lights 271 / 45 (1080p, 64 samples per pixel).
#define RTX_LIGHT_BVH_MIN 96
#endif
constexpr uint32_t kLightsGeneral = 1u;  // SceneView::light_flags: RTX_LIGHTS_GENERAL semantics
constexpr uint32_t kLightsTree = 2u;     // nested lists, chains or XY / YZ rect lights present: tree kernels

// BVH over top-level items (a bare primitive, or one transformed instance such as a Box under
// Rotate / Translate).  A node holds the boxes of BOTH its children, so one 32-byte fetch (two
// 16-byte loads per lane) decides two subtrees; traversal is bound by these fetches, not by
// arithmetic, so the boxes are 16-bit coordinates on a grid over the tree's bounds
// (world = bvh_origin + q * bvh_scale per axis, rounded outwards when built), decoded inside the
// slab test's fused multiply-adds.  A child reference is either the index of another node or,
// with kBvhLeaf set, a leaf: items [ref & kBvhLeafFirst, + ((ref >> 27) & 15)] of `bvh_items`.
// Per child and axis the two planes share one 32-bit word (lo | hi << 16): the fast trace kernel
// turns it into (near | far << 16) for its ray with one rotate by 0 or 16 bits instead of a min and
// a max per axis.
struct alignas(32) BvhNode {
    uint16_t box0[3][2];  // [axis][0 = lo, 1 = hi]
    uint16_t box1[3][2];
    uint32_t child0, child1;
};
// Entries of the traversal's per-lane stack: one deferred child per level, so a tree may be at
// most kBvhStackEntries - 1 levels deep (the builder bounds it, the upload checks it).  The fast
// BVH trace kernel's occupancy hangs on this number: 256 bytes of LDS per wave and entry, 5.5 KB
// per wave as it stands, and 6 KB would cost it two resident waves per CU.
constexpr int kBvhStackEntries = 22;
constexpr uint32_t kBvhLeaf = 0x80000000u;
constexpr uint32_t kBvhLeafFirst = 0x07ffffffu;
constexpr uint32_t kBvhLeafCountShift = 27u;

// Fast-mode leaf record, one per BVH item in leaf order, so that a leaf costs one 32-byte
// fetch instead of the chain item -> steps -> primitive:
//   kBvhItemSphere  f = centre xyz, r*r                       (untransformed sphere)
//   kBvhItemRect*   f = x0 x1 y0 y1 k                         (untransformed rect of that axis)
//   kBvhItemBox     map = world -> unit-cube affine map (BvhMap, rows of the 3x3, then the
//                   offset): a complete box under any Translate / Rotate chain; the scaling of
//                   the map keeps the ray parameter t, so the slab test is against [0,1]^3
//   kBvhItemGroup   map = world -> group affine map; primitives [first, first + count) of `prims`
// `slot` is the item's first position in scan_info / scan_orig (box: + face, group: + index).
struct alignas(32) BvhFastItem {
    float f[5];
    uint32_t slot;
    uint32_t type_count;  // type | count << 8
    uint32_t map;         // box / group: index into BvhMap[]
    // group only: the first primitive travels in the bits of f[0]
};
struct alignas(16) BvhMap {
    float m[12];
};
constexpr uint32_t kBvhItemRectXY = 0u, kBvhItemRectXZ = 1u, kBvhItemRectYZ = 2u;  // = RTX_PRIM_*
constexpr uint32_t kBvhItemSphere = 3u, kBvhItemBox = 4u, kBvhItemGroup = 5u;

struct SceneView {
    const Prim* prims;      // n_prims + 1 records (one readable pad record at the end)
    const PrimInfo* info;   // n_prims records
    const Group* groups;
    const Step* steps;
    const Material* materials;
    const Texture* textures;
    const uint8_t* texels;  // 8-bit RGB image data of all image textures
    const Light* lights;
    // bucketed copy for the fast-mode linear scan (scan_groups / scan_recs null when a BVH is used)
    const ScanGroup* scan_groups;
    const ScanRec* scan_recs;      // one per scan position + 1 readable pad record
    const ScanHit* scan_info;      // in scan order (BVH: in leaf-record slot order)
    const uint32_t* scan_orig;     // scan position -> original primitive index
    uint32_t n_scan_groups;
    uint32_t scan_pad_;
    const BvhNode* bvh;        // null: linear scan over groups
    const Group* bvh_items;    // the groups in leaf order (one load from leaf to primitive range),
                               // then the n_bvh_big groups outside the tree that every ray tests
    const BvhFastItem* bvh_fast;  // the same items as fast-mode leaf records
    const BvhMap* bvh_maps;       // affine maps of the box / group records
    float bvh_origin[3], bvh_scale[3];  // the grid of the nodes' 16-bit boxes
    uint32_t bvh_big_first, n_bvh_big;
    uint32_t n_prims, n_groups, n_steps, n_materials, n_textures, n_lights, n_bvh_nodes;
    uint32_t light_is_list;
    const LightNode* light_nodes;  // n_light_entries records
    uint32_t n_light_entries;      // all entries of lights[] / light_nodes[]; n_lights = the top-level ones
    uint32_t light_flags;          // kLights*
    float cam_origin[3], cam_ll[3], cam_u[3], cam_v[3];
    float background[3];
    uint32_t background_mode;  // RTX_BACKGROUND_*
    float bounds_lo[3], bounds_hi[3];  // padded bounds of all primitives: a ray outside them hits nothing
    const LightBvhNode* light_bvh;  // null: light_pdf_tree visits every leaf
    uint32_t n_light_bvh;
    uint32_t light_bvh_pad_;
    // n_lights as a float and its reciprocal (fast mode): wave-uniform operands from the kernel's
    // arguments, where the conversion and the reciprocal worked out by the kernel were kept in two
    // vector registers across its loop - and spilled
    float n_lights_f, inv_n_lights;
};

struct TraceArgs {
    SceneView scene;
    uint32_t width, height;
    uint32_t spp_begin, spp_count;
    uint32_t max_depth;
    uint32_t accumulate;
    uint32_t tile_w_log2, tile_h_log2;  // pixel tile of one wave: 2x2 .. 8x8
    uint64_t seed;
    float* accum;                 // [height][width][3], row 0 = top
    unsigned long long* rays;     // optional
    float* sample_rgb;            // optional, [pixel][spp_count][3]
    uint32_t* sample_segments;    // optional, [pixel][spp_count]
};

struct HitArgs {
    SceneView scene;
    uint32_t n;
    float t0, t1;
    const float* origins;
    const float* directions;
    void* hits;  // RtxHit[n]
};

}  // namespace rtxd
