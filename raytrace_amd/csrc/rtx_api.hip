// rtx_api.hip — the C ABI of include/rtx.h: context, scene upload (repacking the
// caller's struct-of-arrays into the 32-byte device records of rtx_device.h and
// building the BVH), and the render / hit / resolve entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <array>
#include <vector>

#include "rtx.h"
#include "rtx_bvh.h"
#include "rtx_device.h"
#include "rtx_launch.h"
#include "rtx_special.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& what) {
    g_last_error = what;
    return code;
}

#define RTX_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(RTX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

template <typename T>
struct DeviceArray {
    T* ptr = nullptr;
    size_t count = 0;
    ~DeviceArray() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    hipError_t resize(size_t n) {
        if (n <= count && ptr) return hipSuccess;
        release();
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
        if (e == hipSuccess) count = n;
        return e;
    }
    hipError_t upload(const std::vector<T>& host) {
        hipError_t e = resize(std::max<size_t>(host.size(), 1));
        if (e != hipSuccess || host.empty()) return e;
        return hipMemcpy(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// Device memory of rtx_render_noalloc: the ray counter (first 256 bytes) and the fp32 SUM frame
// behind it.  A file-scope array, so the entry point never allocates; each GPU the process uses
// holds its own copy until the process ends.  In host code the array's name is only a handle:
// the address on a given GPU comes from hipGetSymbolAddress with that GPU current
// (scratch_pointer below), and that pointer is what kernels, memsets and copies get.
constexpr size_t kScratchCounterBytes = 256;
constexpr size_t kScratchBytes = kScratchCounterBytes + (size_t)RTX_NOALLOC_MAX_PIXELS * 3u * sizeof(float);
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};  // per device; written once, under g_scratch_mutex
std::mutex g_scratch_mutex;

// The scratch array's address on `device`, which must be the current device.
hipError_t scratch_pointer(int device, void** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

}  // namespace

struct RtxContext {
    int device = 0;
    bool has_scene = false;
    rtxd::SceneView view{};
    RtxSceneInfo info{};
    DeviceArray<rtxd::Prim> prims;
    DeviceArray<rtxd::PrimInfo> prim_info;
    DeviceArray<rtxd::Group> groups;
    DeviceArray<rtxd::Step> steps;
    DeviceArray<rtxd::Material> materials;
    DeviceArray<rtxd::Texture> textures;
    DeviceArray<uint8_t> texels;
    DeviceArray<rtxd::Light> lights;
    DeviceArray<rtxd::LightNode> light_nodes;
    DeviceArray<rtxd::LightBvhNode> light_bvh;
    DeviceArray<rtxd::ScanGroup> scan_groups;
    DeviceArray<rtxd::ScanRec> scan_recs;
    DeviceArray<rtxd::ScanHit> scan_info;
    DeviceArray<rtxd::BvhFastItem> bvh_fast;
    DeviceArray<rtxd::BvhMap> bvh_maps;
    DeviceArray<uint32_t> scan_orig;
    DeviceArray<rtxd::BvhNode> bvh;
    DeviceArray<rtxd::Group> bvh_items;
    // scratch owned by the host-buffer entry points
    DeviceArray<float> accum;
    DeviceArray<unsigned long long> rays;
    DeviceArray<float> sample_rgb;
    DeviceArray<uint32_t> sample_segments;
    DeviceArray<float> ray_o, ray_d;
    DeviceArray<RtxHit> hits;
    DeviceArray<uint8_t> rgb8;
    DeviceArray<float> peer_frames;  // rtx_render_multi: the other contexts' partial frames, on context 0's GPU
    uint32_t specialize_scan = 1;  // RTX_OPTION_SPECIALIZE_SCAN: 0 never, 1 at upload, 2 in the background
    rtxspecial::Kernel special;    // the kernels compiled for the uploaded scene, if any
    std::string special_note;      // why there are none
    std::shared_ptr<rtxspecial::Job> special_job;  // a compilation in the background, not yet adopted
    uint64_t scene_hash = 0;  // of everything rtx_upload_scene was given (rtx_render_multi compares it)
    void* scratch = nullptr;  // this device's copy of the file-scope scratch array (rtx_render_noalloc)
};

namespace {

int check_scene(const RtxScene* s) {
    if (!s) return fail(RTX_ERR_INVALID_ARGUMENT, "scene is null");
    if (s->abi_version != RTX_ABI_VERSION)
        return fail(RTX_ERR_INVALID_ARGUMENT, "RtxScene.abi_version does not match this library");
    if (s->n_prims == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "scene has no primitives");
    if (s->background_mode > RTX_BACKGROUND_SKY) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown background_mode");
    if ((s->n_lights || s->n_light_entries) && (!s->light_kind || !s->light_a || !s->light_b || !s->light_c || !s->light_d || !s->light_e))
        return fail(RTX_ERR_INVALID_ARGUMENT, "a light array is null");
    if (s->n_chains == 0 || !s->chain_begin)
        return fail(RTX_ERR_INVALID_ARGUMENT, "chain table missing (chain 0 must exist and be empty)");
    if (s->chain_begin[0] != 0 || s->chain_begin[1] != 0)
        return fail(RTX_ERR_INVALID_ARGUMENT, "chain 0 must be the empty chain");
    if (s->chain_begin[s->n_chains] != s->n_steps)
        return fail(RTX_ERR_INVALID_ARGUMENT, "chain_begin[n_chains] != n_steps");
    for (uint32_t c = 0; c < s->n_chains; ++c)
        if (s->chain_begin[c] > s->chain_begin[c + 1])
            return fail(RTX_ERR_INVALID_ARGUMENT, "chain_begin is not monotone");
    if (!s->prim_kind || !s->prim_flip || !s->prim_material || !s->prim_chain || !s->prim_a ||
        !s->prim_b || !s->prim_c || !s->prim_d || !s->prim_e)
        return fail(RTX_ERR_INVALID_ARGUMENT, "a primitive array is null");
    if (s->n_steps && (!s->step_kind || !s->step_x || !s->step_y || !s->step_z || !s->step_w))
        return fail(RTX_ERR_INVALID_ARGUMENT, "a step array is null");
    if (s->n_materials == 0 || !s->mat_kind || !s->mat_texture || !s->mat_param)
        return fail(RTX_ERR_INVALID_ARGUMENT, "material table missing");
    if (s->n_textures && (!s->tex_kind || !s->tex_r || !s->tex_g || !s->tex_b || !s->tex_odd || !s->tex_even ||
                          !s->tex_offset))
        return fail(RTX_ERR_INVALID_ARGUMENT, "a texture array is null");
    if (s->n_texel_bytes && !s->texels) return fail(RTX_ERR_INVALID_ARGUMENT, "texels is null");
    for (uint32_t p = 0; p < s->n_prims; ++p) {
        if (s->prim_kind[p] > RTX_PRIM_SPHERE) return fail(RTX_ERR_INVALID_ARGUMENT, "bad prim_kind");
        if (s->prim_material[p] >= s->n_materials)
            return fail(RTX_ERR_INVALID_ARGUMENT, "prim_material out of range");
        if (s->prim_chain[p] >= s->n_chains) return fail(RTX_ERR_INVALID_ARGUMENT, "prim_chain out of range");
    }
    for (uint32_t k = 0; k < s->n_steps; ++k)
        if (s->step_kind[k] > RTX_STEP_ROTATE) return fail(RTX_ERR_INVALID_ARGUMENT, "bad step_kind");
    for (uint32_t m = 0; m < s->n_materials; ++m) {
        if (s->mat_kind[m] > RTX_MAT_DIFFUSE_LIGHT) return fail(RTX_ERR_INVALID_ARGUMENT, "bad mat_kind");
        if (s->mat_kind[m] != RTX_MAT_DIELECTRIC && s->mat_texture[m] >= s->n_textures)
            return fail(RTX_ERR_INVALID_ARGUMENT, "mat_texture out of range");
    }
    for (uint32_t t = 0; t < s->n_textures; ++t) {
        if (s->tex_kind[t] > RTX_TEX_SKY) return fail(RTX_ERR_INVALID_ARGUMENT, "bad tex_kind");
        if (s->tex_kind[t] == RTX_TEX_IMAGE) {
            const uint64_t need = (uint64_t)s->tex_odd[t] * s->tex_even[t] * 3u;
            if (need == 0 || (uint64_t)s->tex_offset[t] + need > s->n_texel_bytes)
                return fail(RTX_ERR_INVALID_ARGUMENT, "image texture outside the texel buffer");
        }
        if (s->tex_kind[t] == RTX_TEX_CHECKER &&
            (s->tex_odd[t] >= s->n_textures || s->tex_even[t] >= s->n_textures))
            return fail(RTX_ERR_INVALID_ARGUMENT, "checker sub-texture out of range");
    }
    if (s->light_sampling > RTX_LIGHTS_GENERAL) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown light_sampling");
    const uint32_t n_entries = s->n_light_entries ? s->n_light_entries : s->n_lights;
    if (n_entries < s->n_lights) return fail(RTX_ERR_INVALID_ARGUMENT, "n_light_entries < n_lights");
    for (uint32_t l = 0; l < n_entries; ++l) {
        if (s->light_kind[l] > RTX_LIGHT_LIST) return fail(RTX_ERR_INVALID_ARGUMENT, "bad light_kind");
        if (s->light_kind[l] == RTX_LIGHT_LIST) {
            if (!s->light_first || !s->light_count)
                return fail(RTX_ERR_INVALID_ARGUMENT, "a nested light list needs light_first / light_count");
            const uint32_t first = s->light_first[l], count = s->light_count[l];
            // children lie behind their list and behind the top-level entries: no cycles, no sharing
            if (count == 0 || first <= l || first < s->n_lights || first > n_entries || count > n_entries - first)
                return fail(RTX_ERR_INVALID_ARGUMENT, "nested light list: bad child range");
        } else if (s->light_sampling == RTX_LIGHTS_GENERAL && s->light_chain && s->light_chain[l] >= s->n_chains) {
            return fail(RTX_ERR_INVALID_ARGUMENT, "light_chain out of range");
        }
    }
    {   // nesting depth of the light lists (children lie behind their list, so one forward pass)
        std::vector<uint32_t> depth(n_entries, 0u);
        for (uint32_t l = 0; l < n_entries; ++l) {
            if (s->light_kind[l] != RTX_LIGHT_LIST) continue;
            if (depth[l] >= 63u) return fail(RTX_ERR_UNSUPPORTED, "light lists nested deeper than 64 levels");
            for (uint32_t k = 0; k < s->light_count[l]; ++k)
                depth[s->light_first[l] + k] = std::max(depth[s->light_first[l] + k], depth[l] + 1u);
        }
    }
    if (!s->light_is_list && s->n_lights != 1)
        return fail(RTX_ERR_INVALID_ARGUMENT,
                    "a bare (non-list) light shape must have exactly one entry (no light at all: n_lights = 0 "
                    "with light_is_list = 1)");
    return RTX_OK;
}

// True if texture t or any checker sub-texture of it is an image (which reads (u, v)).
bool texture_needs_uv(const RtxScene* s, uint32_t t, int depth) {
    if (depth > 16) return false;
    if (s->tex_kind[t] == RTX_TEX_IMAGE) return true;
    if (s->tex_kind[t] == RTX_TEX_CHECKER)
        return texture_needs_uv(s, s->tex_odd[t], depth + 1) || texture_needs_uv(s, s->tex_even[t], depth + 1);
    return false;
}

// Compose a group's step chain (outermost wrapper first) into one world->group affine map.
// Each step maps p to p - offset (Translate.cpp:7) or to conj(q) p (Rotate.cpp:7-9).
void compose_chain(const std::vector<rtxd::Step>& steps, uint32_t begin, uint32_t end, float out[12]) {
    double M[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, m[3] = {0, 0, 0};
    for (uint32_t s = begin; s < end; ++s) {
        const rtxd::Step& S = steps[s];
        if (S.kind == RTX_STEP_TRANSLATE) {
            m[0] -= S.x; m[1] -= S.y; m[2] -= S.z;
            continue;
        }
        const double x = -S.x, y = -S.y, z = -S.z, w = S.w;  // conjugate
        const double R[3][3] = {
            {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
            {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
            {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
        double M2[3][3], m2[3];
        for (int i = 0; i < 3; ++i) {
            m2[i] = R[i][0] * m[0] + R[i][1] * m[1] + R[i][2] * m[2];
            for (int j = 0; j < 3; ++j) M2[i][j] = R[i][0] * M[0][j] + R[i][1] * M[1][j] + R[i][2] * M[2][j];
        }
        std::memcpy(M, M2, sizeof M);
        std::memcpy(m, m2, sizeof m);
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out[3 * i + j] = (float)M[i][j];
        out[9 + i] = (float)m[i];
    }
}

// The ScanHit of one primitive of group g (world->group map S.m).
rtxd::ScanHit make_scan_hit(const rtxd::Prim& P, const rtxd::PrimInfo& I, const rtxd::ScanGroup& S, uint32_t g) {
    rtxd::ScanHit H{};
    H.r = I.r; H.g = I.g; H.b = I.b; H.param = I.param;
    H.bits = I.bits; H.texture = I.texture;
    if (((I.bits >> rtxd::kInfoMatShift) & 3u) == RTX_MAT_DIELECTRIC) {
        // a Dielectric has no colour (its weight is 1, Dielectric.cpp:22): the slots carry what its
        // scattering code would otherwise work out per hit, 1 / ri and Schlick's r0 (Util.h:73-76)
        const double ri = I.param, q = (1.0 - ri) / (1.0 + ri);
        H.r = (float)(1.0 / ri);
        H.g = (float)(q * q);
    }
    H.group = g;
    H.chained = S.identity ? 0u : 1u;
    // group -> world: p_world = M^T (p_local - m)
    const uint32_t kind = P.kind_flip & 3u;
    const float sign = (P.kind_flip & 4u) ? -1.0f : 1.0f;
    if (kind == RTX_PRIM_SPHERE) {
        const double l[3] = {(double)P.a - S.m[9], (double)P.b - S.m[10], (double)P.c - S.m[11]};
        for (int j = 0; j < 3; ++j) H.c[j] = (float)(S.m[0 + j] * l[0] + S.m[3 + j] * l[1] + S.m[6 + j] * l[2]);
        H.inv_r = sign / P.d;
    } else {
        const int axis = kind == RTX_PRIM_RECT_XY ? 2 : (kind == RTX_PRIM_RECT_XZ ? 1 : 0);
        for (int j = 0; j < 3; ++j) H.n0[j] = sign * S.m[3 * axis + j];
        H.inv_r = 0.0f;
    }
    return H;
}

// Rects of [begin, end) that are the six complete faces of one axis-aligned box: returns the
// boxes as primitive indices in face order +z, -z, +y, -y, +x, -x and marks them used.
std::vector<std::array<uint32_t, 6>> find_boxes(const std::vector<rtxd::Prim>& prims, uint32_t begin, uint32_t end,
                                                 std::vector<char>& used) {
    std::vector<std::array<uint32_t, 6>> boxes;
    auto kind = [&](uint32_t p) { return prims[p].kind_flip & 3u; };
    auto face = [&](uint32_t want_kind, float x0, float x1, float y0, float y1, float k) -> int {
        for (uint32_t p = begin; p < end; ++p) {
            const rtxd::Prim& P = prims[p];
            if (!used[p] && kind(p) == want_kind && P.a == x0 && P.b == x1 && P.c == y0 && P.d == y1 && P.e == k)
                return (int)p;
        }
        return -1;
    };
    for (uint32_t p = begin; p < end; ++p) {
        if (used[p] || kind(p) != RTX_PRIM_RECT_XY) continue;
        for (uint32_t q = p + 1; q < end; ++q) {
            if (used[q] || kind(q) != RTX_PRIM_RECT_XY) continue;
            const rtxd::Prim &A = prims[p], &B = prims[q];
            if (A.a != B.a || A.b != B.b || A.c != B.c || A.d != B.d || A.e == B.e) continue;
            if (!(A.a < A.b) || !(A.c < A.d)) continue;
            const float X0 = A.a, X1 = A.b, Y0 = A.c, Y1 = A.d, Z0 = std::min(A.e, B.e), Z1 = std::max(A.e, B.e);
            used[p] = used[q] = 1;  // so that face() skips them
            const int zh = A.e == Z1 ? (int)p : (int)q, zl = A.e == Z1 ? (int)q : (int)p;
            const int yh = face(RTX_PRIM_RECT_XZ, X0, X1, Z0, Z1, Y1);
            if (yh >= 0) used[yh] = 1;
            const int yl = face(RTX_PRIM_RECT_XZ, X0, X1, Z0, Z1, Y0);
            if (yl >= 0) used[yl] = 1;
            const int xh = face(RTX_PRIM_RECT_YZ, Y0, Y1, Z0, Z1, X1);
            if (xh >= 0) used[xh] = 1;
            const int xl = face(RTX_PRIM_RECT_YZ, Y0, Y1, Z0, Z1, X0);
            if (xl >= 0) used[xl] = 1;
            if (yh >= 0 && yl >= 0 && xh >= 0 && xl >= 0) {
                boxes.push_back({(uint32_t)zh, (uint32_t)zl, (uint32_t)yh, (uint32_t)yl, (uint32_t)xh, (uint32_t)xl});
                break;
            }
            for (int f : {(int)p, (int)q, yh, yl, xh, xl})
                if (f >= 0) used[f] = 0;
        }
    }
    return boxes;
}

void build_scan_table(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::PrimInfo>& info,
                      const std::vector<rtxd::Group>& groups, const std::vector<rtxd::Step>& steps,
                      std::vector<rtxd::ScanGroup>& sg, std::vector<rtxd::ScanRec>& sr,
                      std::vector<rtxd::ScanHit>& si, std::vector<uint32_t>& so) {
    std::vector<char> used(prims.size(), 0);
    for (uint32_t g = 0; g < groups.size(); ++g) {
        const rtxd::Group& G = groups[g];
        rtxd::ScanGroup S{};
        compose_chain(steps, G.step_begin, G.step_end, S.m);
        S.identity = G.step_end == G.step_begin ? 1u : 0u;
        // Rotate about (0,1,0) - what the reference's own scene uses (Scene.cpp:75-77) - composes to a
        // matrix whose y row and column are those of the identity, exactly
        S.rot_y = (!S.identity && S.m[1] == 0.0f && S.m[3] == 0.0f && S.m[5] == 0.0f && S.m[7] == 0.0f && S.m[4] == 1.0f) ? 1u : 0u;
        S.first = (uint32_t)sr.size();
        const auto boxes = find_boxes(prims, G.prim_begin, G.prim_end, used);
        uint32_t* counts[4] = {&S.n_xy, &S.n_xz, &S.n_yz, &S.n_sph};
        for (uint32_t kind = 0; kind < 4; ++kind) {
            for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) {
                const rtxd::Prim& P = prims[p];
                if (used[p] || (P.kind_flip & 3u) != kind) continue;
                rtxd::ScanRec R{};
                R.a = P.a; R.b = P.b; R.c = P.c; R.d = P.d;
                R.e = kind == RTX_PRIM_SPHERE ? P.d * P.d : P.e;
                sr.push_back(R);
                si.push_back(make_scan_hit(P, info[p], S, g));
                so.push_back(p);
                ++*counts[kind];
            }
        }
        for (const auto& box : boxes) {
            const rtxd::Prim &ZH = prims[box[0]], &ZL = prims[box[1]], &YH = prims[box[2]], &YL = prims[box[3]];
            rtxd::ScanRec R{};
            R.a = ZH.a; R.b = ZH.c; R.c = ZL.e;  // lo
            R.d = ZH.b; R.e = ZH.d; R.f = ZH.e;  // hi
            (void)YH; (void)YL;
            for (uint32_t f = 0; f < rtxd::kBoxSlots; ++f) {
                sr.push_back(f == 0 ? R : rtxd::ScanRec{});
                si.push_back(make_scan_hit(prims[box[f]], info[box[f]], S, g));
                so.push_back(box[f]);
            }
            ++S.n_box;
        }
        sg.push_back(S);
    }
    sr.push_back(sr.back());  // readable pad record
}

// Fast-mode leaf records of the BVH items (groups in leaf order) and the hit-completion table
// in their slot order.
void build_bvh_fast_items(const std::vector<rtxd::Prim>& prims, const std::vector<rtxd::PrimInfo>& info,
                          const std::vector<rtxd::Group>& items, const std::vector<rtxd::Step>& steps,
                          std::vector<rtxd::BvhFastItem>& out, std::vector<rtxd::BvhMap>& maps,
                          std::vector<rtxd::ScanHit>& si, std::vector<uint32_t>& so) {
    std::vector<char> used(prims.size(), 0);
    for (uint32_t g = 0; g < items.size(); ++g) {
        const rtxd::Group& G = items[g];
        rtxd::ScanGroup S{};
        compose_chain(steps, G.step_begin, G.step_end, S.m);
        S.identity = G.step_end == G.step_begin ? 1u : 0u;
        rtxd::BvhFastItem I{};
        I.slot = (uint32_t)si.size();
        const uint32_t n = G.prim_end - G.prim_begin;
        const uint32_t group_of_prims = prims[G.prim_begin].group;
        auto push = [&](uint32_t p) {
            si.push_back(make_scan_hit(prims[p], info[p], S, group_of_prims));
            so.push_back(p);
        };
        std::vector<std::array<uint32_t, 6>> boxes;
        if (n == 6) boxes = find_boxes(prims, G.prim_begin, G.prim_end, used);
        if (n == 1 && S.identity) {
            const rtxd::Prim& P = prims[G.prim_begin];
            const uint32_t type = P.kind_flip & 3u;  // kBvhItemRect* / kBvhItemSphere = RTX_PRIM_*
            I.type_count = type | (1u << 8);
            I.f[0] = P.a; I.f[1] = P.b; I.f[2] = P.c;
            I.f[3] = type == rtxd::kBvhItemSphere ? P.d * P.d : P.d;
            I.f[4] = P.e;
            push(G.prim_begin);
        } else if (boxes.size() == 1) {
            const auto& box = boxes[0];
            const rtxd::Prim &ZH = prims[box[0]], &ZL = prims[box[1]];
            const double lo[3] = {ZH.a, ZH.c, ZL.e}, hi[3] = {ZH.b, ZH.d, ZH.e};
            rtxd::BvhMap M{};
            for (int i = 0; i < 3; ++i) {
                const double inv = 1.0 / (hi[i] - lo[i]);
                for (int j = 0; j < 3; ++j) M.m[3 * i + j] = (float)(S.m[3 * i + j] * inv);
                M.m[9 + i] = (float)((S.m[9 + i] - lo[i]) * inv);
            }
            I.map = (uint32_t)maps.size();
            maps.push_back(M);
            I.type_count = rtxd::kBvhItemBox | (6u << 8);
            for (uint32_t f = 0; f < 6; ++f) push(box[f]);
        } else {
            for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) used[p] = 0;  // a partial box match
            rtxd::BvhMap M{};
            std::memcpy(M.m, S.m, sizeof M.m);
            I.map = (uint32_t)maps.size();
            maps.push_back(M);
            I.type_count = rtxd::kBvhItemGroup | (n << 8);
            std::memcpy(&I.f[0], &G.prim_begin, sizeof(uint32_t));  // the first primitive, as bits
            for (uint32_t p = G.prim_begin; p < G.prim_end; ++p) push(p);
        }
        out.push_back(I);
    }
}

// One record per primitive with all the integrator needs after a hit.
std::vector<rtxd::PrimInfo> prim_infos(const RtxScene* s) {
    std::vector<rtxd::PrimInfo> info(s->n_prims);
    for (uint32_t p = 0; p < s->n_prims; ++p) {
        rtxd::PrimInfo& I = info[p];
        const uint32_t m = s->prim_material[p];
        const uint32_t mk = s->mat_kind[m];
        I.param = s->mat_param[m];
        I.bits = s->prim_kind[p] | (s->prim_flip[p] ? rtxd::kInfoFlip : 0u) | (mk << rtxd::kInfoMatShift);
        I.texture = 0;
        I.r = I.g = I.b = 1.0f;
        if (mk != RTX_MAT_DIELECTRIC) {
            const uint32_t t = s->mat_texture[m];
            I.texture = t;
            if (s->tex_kind[t] == RTX_TEX_COLOR) {
                I.r = s->tex_r[t]; I.g = s->tex_g[t]; I.b = s->tex_b[t];
            } else {
                I.bits |= rtxd::kInfoTextured;
                if (texture_needs_uv(s, t, 0)) I.bits |= rtxd::kInfoNeedsUV;
            }
        }
        const uint32_t c = s->prim_chain[p];
        I.step_begin = s->chain_begin[c];
        I.step_end = s->chain_begin[c + 1];
    }
    return info;
}

// Whether the scene's light sampling needs the kernels that walk a light tree (nested lists, lights
// under transforms, enough leaves for a tree over them): what rtx_upload_scene finds while it builds
// the light tables, from the scene alone.
bool lights_need_tree(const RtxScene* s) {
    const uint32_t n_entries = s->n_light_entries ? s->n_light_entries : s->n_lights;
    const bool general = s->light_sampling == RTX_LIGHTS_GENERAL;
    size_t with_density = 0;
    for (uint32_t l = 0; l < n_entries; ++l) {
        const uint32_t kind = s->light_kind[l];
        if (kind == RTX_LIGHT_LIST) return true;
        if (general) {
            const uint32_t c = s->light_chain ? s->light_chain[l] : 0u;
            if (s->chain_begin[c + 1] > s->chain_begin[c] || kind == RTX_LIGHT_RECT_XY || kind == RTX_LIGHT_RECT_YZ) return true;
        }
        if (kind == RTX_LIGHT_SPHERE || kind == RTX_LIGHT_RECT_XZ || (general && kind <= RTX_LIGHT_RECT_YZ)) ++with_density;
    }
    return with_density >= RTX_LIGHT_BVH_MIN;
}

// Entry l of the scene's light arrays as the kernels read it.
rtxd::Light light_record(const RtxScene* s, uint32_t l) {
    rtxd::Light E;
    E.a = s->light_a[l]; E.b = s->light_b[l]; E.c = s->light_c[l]; E.d = s->light_d[l]; E.e = s->light_e[l];
    E.kind = s->light_kind[l];
    E.inv_area = E.kind <= RTX_LIGHT_RECT_YZ ? 1.0f / ((E.b - E.a) * (E.d - E.c)) : 0.0f;
    E.pad = 0;
    return E;
}

// The text of the kernels compiled for a scan scene: the scan and, where the density sum of a diffuse
// bounce runs over a plain list (no light tree), the light entries.
std::string special_text(const RtxScene* s, bool light_tree, const std::vector<rtxd::ScanGroup>& scan_groups,
                         const std::vector<rtxd::ScanRec>& scan_recs, const std::vector<rtxd::ScanHit>& scan_hits) {
    std::vector<rtxd::Light> lights(s->n_lights);
    for (uint32_t l = 0; l < s->n_lights; ++l) lights[l] = light_record(s, l);
    std::string text = rtxspecial::features_source(scan_hits, lights, s->n_lights, light_tree);
    text += rtxspecial::scan_source(scan_groups, scan_recs);
    if (!light_tree && s->n_lights != 0) text += rtxspecial::lights_source(lights, s->n_lights, s->light_is_list != 0);
    return text;
}

// The trace kernel variant rtx_fast_mode::launch_trace picks for the frames of a scan scene.
rtxspecial::Variant scan_variant(const RtxScene* s, bool light_tree) {
    rtxspecial::Variant v;
    v.light_tree = light_tree;
    v.list_lit = !light_tree && s->light_is_list && s->n_lights != 0 && s->background_mode != RTX_BACKGROUND_SKY;
    return v;
}

// Takes over the kernels a background compilation (the helper program's, see rtx_special.h) has finished -
// called by the render entry points, with the context's device current: from the process's cache onto this device.
void adopt_special_job(RtxContext* ctx) {
    if (!ctx->special_job) return;
    const int state = ctx->special_job->state.load(std::memory_order_acquire);
    if (state == 0) return;
    const std::shared_ptr<rtxspecial::Job> job = std::move(ctx->special_job);
    ctx->special_job.reset();
    rtxspecial::finish(job);  // (the thread has set its state: this returns at once)
    if (state == 1 && rtxspecial::load(job->text, job->variant, ctx->special, ctx->special_note, nullptr)) {
        ctx->special_note.clear();
        ctx->info.scan_specialized = 1;
        ctx->info.specialize_seconds = (float)job->seconds;
    } else if (state == 2) {
        ctx->special_note = job->why;
    }
}

// A trace kernel compiled for the uploaded scene: the launch rtx_fast_mode::launch_trace makes for a
// scan scene, through the module API.
hipError_t launch_special_trace(hipFunction_t kernel, rtxd::TraceArgs a, hipStream_t stream) {
    rtxd::choose_tile(a.spp_count, a.width, a.height, false, a.tile_w_log2, a.tile_h_log2);
    const uint32_t bw = 2u << a.tile_w_log2, bh = 2u << a.tile_h_log2;  // 2x2 waves per 256-thread block
    const uint32_t blocks = ((a.width + bw - 1u) / bw) * ((a.height + bh - 1u) / bh);
    size_t size = sizeof a;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(kernel, blocks, 1, 1, 256, 1, 1, 0, stream, nullptr, config);
}
hipError_t launch_special_hit(hipFunction_t kernel, rtxd::HitArgs a, hipStream_t stream) {
    size_t size = sizeof a;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(kernel, (a.n + 255u) / 256u, 1, 1, 256, 1, 1, 0, stream, nullptr, config);
}

// FNV-1a over the bytes of the caller's scene: contexts that are to render one frame together
// must have been given the same arrays.
struct SceneHasher {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    }
    template <typename T>
    void array(const T* p, size_t n) {
        const uint64_t tag = n;
        bytes(&tag, sizeof tag);
        if (p && n) bytes(p, n * sizeof(T));
    }
};
uint64_t hash_scene(const RtxScene* s, bool use_bvh) {
    SceneHasher H;
    const uint32_t n_le = s->n_light_entries ? s->n_light_entries : s->n_lights;
    const uint32_t head[] = {s->n_prims, s->n_chains, s->n_steps, s->n_materials, s->n_textures, s->n_lights,
                             s->light_is_list, n_le, s->light_sampling, s->background_mode, use_bvh ? 1u : 0u};
    H.array(head, sizeof head / sizeof head[0]);
    H.array(s->prim_kind, s->n_prims); H.array(s->prim_flip, s->n_prims); H.array(s->prim_material, s->n_prims);
    H.array(s->prim_chain, s->n_prims);
    H.array(s->prim_a, s->n_prims); H.array(s->prim_b, s->n_prims); H.array(s->prim_c, s->n_prims);
    H.array(s->prim_d, s->n_prims); H.array(s->prim_e, s->n_prims);
    H.array(s->chain_begin, (size_t)s->n_chains + 1u);
    H.array(s->step_kind, s->n_steps);
    H.array(s->step_x, s->n_steps); H.array(s->step_y, s->n_steps); H.array(s->step_z, s->n_steps);
    H.array(s->step_w, s->n_steps);
    H.array(s->mat_kind, s->n_materials); H.array(s->mat_texture, s->n_materials); H.array(s->mat_param, s->n_materials);
    H.array(s->tex_kind, s->n_textures);
    H.array(s->tex_r, s->n_textures); H.array(s->tex_g, s->n_textures); H.array(s->tex_b, s->n_textures);
    H.array(s->tex_odd, s->n_textures); H.array(s->tex_even, s->n_textures); H.array(s->tex_offset, s->n_textures);
    H.array(s->texels, (size_t)s->n_texel_bytes);
    H.array(s->light_kind, n_le);
    H.array(s->light_a, n_le); H.array(s->light_b, n_le); H.array(s->light_c, n_le); H.array(s->light_d, n_le);
    H.array(s->light_e, n_le);
    H.array(s->light_first, s->light_first ? n_le : 0u); H.array(s->light_count, s->light_count ? n_le : 0u);
    H.array(s->light_chain, s->light_chain ? n_le : 0u);
    H.array(s->cam_origin, 3); H.array(s->cam_ll, 3); H.array(s->cam_u, 3); H.array(s->cam_v, 3);
    H.array(s->background, 3);
    return H.h;
}

int check_params(const RtxContext* ctx, const RtxParams* p) {
    if (!ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "context is null");
    if (!p) return fail(RTX_ERR_INVALID_ARGUMENT, "params is null");
    if (!ctx->has_scene) return fail(RTX_ERR_NO_SCENE, "no scene uploaded");
    if (p->width == 0 || p->height == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "empty image");
    if ((uint64_t)p->width * p->height > (1ull << 30))
        return fail(RTX_ERR_TOO_LARGE, "more than 2^30 pixels");
    if (p->mode > RTX_MODE_DETERMINISTIC) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown mode");
    if ((uint64_t)p->spp_begin + p->spp_count > 0xffffffffull)
        return fail(RTX_ERR_INVALID_ARGUMENT, "sample range overflows 32 bits");
    if (p->spp_count > (1u << 28))  // a wave's item counter (pixels of its tile x samples) is 32 bits
        return fail(RTX_ERR_TOO_LARGE, "more than 2^28 samples per pixel in one call: split the range");
    return RTX_OK;
}

rtxd::TraceArgs make_trace_args(const RtxContext* ctx, const RtxParams* p) {
    rtxd::TraceArgs a{};
    a.scene = ctx->view;
    a.width = p->width;
    a.height = p->height;
    a.spp_begin = p->spp_begin;
    a.spp_count = p->spp_count;
    a.max_depth = p->max_depth;
    a.accumulate = p->accumulate;
    a.seed = p->seed;
    return a;
}

}  // namespace

extern "C" {

const char* rtx_last_error(void) { return g_last_error.c_str(); }

int rtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int rtx_create(int device, RtxContext** out_ctx) {
    if (!out_ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return fail(RTX_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");
    }
    if (device < 0 || device >= n || device >= kMaxDevices)
        return fail(RTX_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    RTX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RTX_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(RTX_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    RtxContext* ctx = new (std::nothrow) RtxContext();
    if (!ctx) return fail(RTX_ERR_HIP, "out of host memory");
    ctx->device = device;
    hipError_t e = scratch_pointer(device, &ctx->scratch);  // once per device, with that device current
    if (e != hipSuccess) {
        delete ctx;
        return fail(RTX_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    }
    *out_ctx = ctx;
    return RTX_OK;
}

void rtx_destroy(RtxContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    rtxspecial::finish(ctx->special_job);  // (ends a compilation still running)
    rtxspecial::unload(ctx->special);
    delete ctx;
}

int rtx_set_option(RtxContext* ctx, uint32_t option, uint64_t value) {
    if (!ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "context is null");
    if (option == RTX_OPTION_SPECIALIZE_SCAN && value <= 2) {
        ctx->specialize_scan = (uint32_t)value;
        return RTX_OK;
    }
    return fail(RTX_ERR_INVALID_ARGUMENT, "unknown option or value");
}

// The scan text of a scene rtx_upload_scene would specialise (LINEAR or AUTO accel), or an error.
static int special_scan_text(const RtxScene* s, std::string& scan) {
    int rc = check_scene(s);
    if (rc != RTX_OK) return rc;
    if (s->n_prims > 64) return fail(RTX_ERR_UNSUPPORTED, "not a scene whose scan is specialised (more than 64 primitives: rendered through the tree)");
    std::vector<rtxd::Prim> prims;
    std::vector<rtxd::Step> steps;
    std::vector<rtxd::Group> groups;
    std::string err;
    if (!rtxbvh::scene_groups(s, false, prims, steps, groups, err)) return fail(RTX_ERR_TOO_LARGE, err);
    std::vector<rtxd::ScanGroup> scan_groups;
    std::vector<rtxd::ScanRec> scan_recs;
    std::vector<rtxd::ScanHit> scan_info;
    std::vector<uint32_t> scan_orig;
    build_scan_table(prims, prim_infos(s), groups, steps, scan_groups, scan_recs, scan_info, scan_orig);
    if (scan_recs.size() - 1 > rtxspecial::kMaxRecords) return fail(RTX_ERR_UNSUPPORTED, "more scan records than a specialised kernel takes");
    scan = special_text(s, lights_need_tree(s), scan_groups, scan_recs, scan_info);
    return RTX_OK;
}

int rtx_prepare_scene(const RtxScene* s) {
    std::string scan, err;
    int rc = special_scan_text(s, scan);
    if (rc != RTX_OK) return rc;
    if (!rtxspecial::prepare(scan, scan_variant(s, lights_need_tree(s)), err, nullptr)) return fail(RTX_ERR_UNSUPPORTED, err);
    return RTX_OK;
}

int rtx_scene_kernel_source(const RtxScene* s, char* text, uint64_t capacity, uint64_t* needed) {
    std::string scan;
    int rc = special_scan_text(s, scan);
    if (rc != RTX_OK) return rc;
    if (needed) *needed = scan.size() + 1;
    if (text && capacity) {
        const size_t n = scan.size() < capacity - 1 ? scan.size() : capacity - 1;
        std::memcpy(text, scan.data(), n);
        text[n] = '\0';
    }
    return RTX_OK;
}

const char* rtx_scene_note(RtxContext* ctx) { return ctx ? ctx->special_note.c_str() : ""; }

int rtx_upload_scene(RtxContext* ctx, const RtxScene* s, uint32_t accel) {
    if (!ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "context is null");
    if (accel > RTX_ACCEL_BVH) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown accel");
    int rc = check_scene(s);
    if (rc != RTX_OK) return rc;
    RTX_HIP(hipSetDevice(ctx->device));
    ctx->has_scene = false;

    const bool use_bvh = accel == RTX_ACCEL_BVH || (accel == RTX_ACCEL_AUTO && s->n_prims > 64);

    // primitives, transform steps and the groups the scan walks / the tree is built over
    std::vector<rtxd::Prim> prims;
    std::vector<rtxd::Step> steps;
    std::vector<rtxd::Group> groups;
    {
        std::string err;
        if (!rtxbvh::scene_groups(s, use_bvh, prims, steps, groups, err)) return fail(RTX_ERR_TOO_LARGE, err);
    }
    std::vector<rtxd::Material> mats(s->n_materials);
    for (uint32_t m = 0; m < s->n_materials; ++m)
        mats[m] = rtxd::Material{s->mat_kind[m], s->mat_texture[m], s->mat_param[m], 0u};
    std::vector<rtxd::Texture> texs(s->n_textures);
    for (uint32_t t = 0; t < s->n_textures; ++t) {
        rtxd::Texture& T = texs[t];
        T.kind = s->tex_kind[t];
        T.r = s->tex_r[t]; T.g = s->tex_g[t]; T.b = s->tex_b[t];
        T.odd = s->tex_odd[t]; T.even = s->tex_even[t];
        T.offset = s->tex_offset[t];
        T.pad = 0;
    }
    const uint32_t n_light_entries = s->n_light_entries ? s->n_light_entries : s->n_lights;
    const bool general_lights = s->light_sampling == RTX_LIGHTS_GENERAL;
    std::vector<rtxd::Light> lights(n_light_entries);
    std::vector<rtxd::LightNode> light_nodes(n_light_entries);
    bool light_tree = false;
    for (uint32_t l = 0; l < n_light_entries; ++l) {
        rtxd::Light& E = lights[l];
        E = light_record(s, l);
        rtxd::LightNode& N = light_nodes[l];  // value-initialised; a nested entry's weight is already set by its list
        // ShapeList::pdf_value weighs each entry by 1.0f / size (ShapeList.cpp:20); nested lists multiply
        if (l < s->n_lights) N.weight = s->light_is_list ? 1.0f / (float)s->n_lights : 1.0f;
        if (E.kind == RTX_LIGHT_LIST) {
            light_tree = true;
            N.first = s->light_first[l];
            N.count = s->light_count[l];
            const float w = 1.0f / (float)N.count;
            for (uint32_t k = 0; k < N.count; ++k) light_nodes[N.first + k].weight = N.weight * w;  // children lie behind l
        } else if (general_lights) {
            const uint32_t c = s->light_chain ? s->light_chain[l] : 0u;
            N.step_begin = s->chain_begin[c];
            N.step_end = s->chain_begin[c + 1];
            if (N.step_end > N.step_begin || E.kind == RTX_LIGHT_RECT_XY || E.kind == RTX_LIGHT_RECT_YZ) light_tree = true;
        }
    }
    // many light leaves with a density of their own: a tree over them, so that the density sum of
    // a diffuse bounce visits the leaves the sampled direction meets instead of all of them
    std::vector<rtxd::LightBvhNode> light_bvh;
    {
        std::vector<rtxd::Prim> shapes;
        std::vector<rtxd::Group> chains;
        std::vector<uint32_t> entries;
        for (uint32_t l = 0; l < n_light_entries; ++l) {
            const rtxd::Light& E = lights[l];
            const bool has_density = E.kind == RTX_LIGHT_SPHERE || E.kind == RTX_LIGHT_RECT_XZ ||
                                     (general_lights && E.kind <= RTX_LIGHT_RECT_YZ);
            if (!has_density) continue;
            rtxd::Prim P{};
            P.a = E.a; P.b = E.b; P.c = E.c; P.d = E.d; P.e = E.e;
            P.kind_flip = E.kind;  // RTX_LIGHT_RECT_* / SPHERE = RTX_PRIM_*
            rtxd::Group G{};
            G.prim_begin = (uint32_t)shapes.size();
            G.prim_end = G.prim_begin + 1u;
            G.step_begin = light_nodes[l].step_begin;
            G.step_end = light_nodes[l].step_end;
            shapes.push_back(P);
            chains.push_back(G);
            entries.push_back(l);
        }
        if (entries.size() >= RTX_LIGHT_BVH_MIN) {
            rtxbvh::build_light_tree(shapes, chains, steps, entries, light_bvh);
            light_tree = true;  // the kernels that know the tree
        }
    }
    const std::vector<rtxd::PrimInfo> info = prim_infos(s);
    // the scan requests record p+1 while it tests record p: keep one readable pad record
    std::vector<rtxd::Prim> prims_padded = prims;
    prims_padded.push_back(prims.back());
    // bucketed table for the fast-mode linear scan
    std::vector<rtxd::ScanGroup> scan_groups;
    std::vector<rtxd::ScanRec> scan_recs;
    std::vector<rtxd::ScanHit> scan_info;
    std::vector<uint32_t> scan_orig;
    if (!use_bvh) build_scan_table(prims, info, groups, steps, scan_groups, scan_recs, scan_info, scan_orig);
    rtxbvh::Tree tree;
    std::vector<rtxd::BvhNode>& nodes = tree.nodes;
    std::vector<rtxd::Group>& items = tree.items;  // leaf order, then the items outside the tree
    std::vector<rtxd::BvhFastItem> fast_items;
    std::vector<rtxd::BvhMap> fast_maps;
    if (use_bvh) {
        std::string err;
        if (!rtxbvh::build(prims, groups, steps, tree, err)) return fail(RTX_ERR_TOO_LARGE, err);
        // a lane defers at most one child per level: the stack must hold the tree's depth
        if (tree.depth + 1 > rtxd::kBvhStackEntries)
            return fail(RTX_ERR_TOO_LARGE, "BVH is deeper than the traversal stack");
        build_bvh_fast_items(prims, info, items, steps, fast_items, fast_maps, scan_info, scan_orig);
    }

    // the kernels address these tables per lane with 32-bit byte offsets
    if (scan_info.size() * sizeof(rtxd::ScanHit) >= (1ull << 32) || fast_items.size() * sizeof(rtxd::BvhFastItem) >= (1ull << 32) ||
        nodes.size() * sizeof(rtxd::BvhNode) >= (1ull << 32) ||
        lights.size() * sizeof(rtxd::Light) >= (1ull << 32) || fast_maps.size() * sizeof(rtxd::BvhMap) >= (1ull << 32) ||
        prims_padded.size() * sizeof(rtxd::Prim) >= (1ull << 32))
        return fail(RTX_ERR_TOO_LARGE, "a scene table exceeds 4 GiB");
    RTX_HIP(ctx->prims.upload(prims_padded));
    RTX_HIP(ctx->prim_info.upload(info));
    RTX_HIP(ctx->groups.upload(groups));
    RTX_HIP(ctx->steps.upload(steps));
    RTX_HIP(ctx->materials.upload(mats));
    RTX_HIP(ctx->textures.upload(texs));
    RTX_HIP(ctx->texels.upload(std::vector<uint8_t>(s->texels, s->texels + s->n_texel_bytes)));
    RTX_HIP(ctx->lights.upload(lights));
    RTX_HIP(ctx->light_nodes.upload(light_nodes));
    RTX_HIP(ctx->light_bvh.upload(light_bvh));
    RTX_HIP(ctx->scan_groups.upload(scan_groups));
    RTX_HIP(ctx->scan_recs.upload(scan_recs));
    RTX_HIP(ctx->scan_info.upload(scan_info));
    RTX_HIP(ctx->scan_orig.upload(scan_orig));
    RTX_HIP(ctx->bvh.upload(nodes));
    RTX_HIP(ctx->bvh_items.upload(items));
    RTX_HIP(ctx->bvh_fast.upload(fast_items));
    RTX_HIP(ctx->bvh_maps.upload(fast_maps));

    rtxd::SceneView& v = ctx->view;
    v.prims = ctx->prims.ptr;
    v.info = ctx->prim_info.ptr;
    v.groups = ctx->groups.ptr;
    v.steps = ctx->steps.ptr;
    v.materials = ctx->materials.ptr;
    v.textures = ctx->textures.ptr;
    v.texels = ctx->texels.ptr;
    v.lights = ctx->lights.ptr;
    v.scan_groups = use_bvh ? nullptr : ctx->scan_groups.ptr;
    v.scan_recs = use_bvh ? nullptr : ctx->scan_recs.ptr;
    v.scan_info = ctx->scan_info.ptr;
    v.scan_orig = ctx->scan_orig.ptr;
    v.n_scan_groups = (uint32_t)scan_groups.size();
    v.scan_pad_ = 0;
    v.bvh = use_bvh ? ctx->bvh.ptr : nullptr;
    v.bvh_items = use_bvh ? ctx->bvh_items.ptr : nullptr;
    v.bvh_fast = use_bvh ? ctx->bvh_fast.ptr : nullptr;
    v.bvh_maps = use_bvh ? ctx->bvh_maps.ptr : nullptr;
    std::memcpy(v.bvh_origin, tree.origin, sizeof v.bvh_origin);
    std::memcpy(v.bvh_scale, tree.scale, sizeof v.bvh_scale);
    v.n_bvh_big = tree.n_big;
    v.bvh_big_first = (uint32_t)items.size() - tree.n_big;
    v.n_prims = s->n_prims;
    v.n_groups = (uint32_t)groups.size();
    v.n_steps = s->n_steps;
    v.n_materials = s->n_materials;
    v.n_textures = s->n_textures;
    v.n_lights = s->n_lights;
    v.n_bvh_nodes = (uint32_t)nodes.size();
    v.light_is_list = s->light_is_list ? 1u : 0u;
    v.light_nodes = ctx->light_nodes.ptr;
    v.n_light_entries = n_light_entries;
    v.light_flags = (general_lights ? rtxd::kLightsGeneral : 0u) | (light_tree ? rtxd::kLightsTree : 0u);
    v.light_bvh = light_bvh.empty() ? nullptr : ctx->light_bvh.ptr;
    v.n_light_bvh = (uint32_t)light_bvh.size();
    v.light_bvh_pad_ = 0;
    v.n_lights_f = (float)s->n_lights;
    v.inv_n_lights = s->n_lights ? 1.0f / (float)s->n_lights : 0.0f;
    std::memcpy(v.cam_origin, s->cam_origin, sizeof v.cam_origin);
    std::memcpy(v.cam_ll, s->cam_ll, sizeof v.cam_ll);
    std::memcpy(v.cam_u, s->cam_u, sizeof v.cam_u);
    std::memcpy(v.cam_v, s->cam_v, sizeof v.cam_v);
    std::memcpy(v.background, s->background, sizeof v.background);
    v.background_mode = s->background_mode;
    rtxbvh::scene_bounds(prims, groups, steps, v.bounds_lo, v.bounds_hi);

    ctx->info.n_prims = s->n_prims;
    ctx->info.n_items = (uint32_t)groups.size();
    ctx->info.n_bvh_nodes = (uint32_t)nodes.size();
    ctx->info.accel = use_bvh ? RTX_ACCEL_BVH : RTX_ACCEL_LINEAR;
    ctx->info.bvh_depth = use_bvh ? (uint32_t)tree.depth : 0u;
    ctx->info.bvh_visits = use_bvh ? (float)tree.visits : 0.0f;
    // the scan of a small scene compiled into the fast-mode trace kernel (the variant launch_trace would pick
    // for this scene's frames)
    rtxspecial::unload(ctx->special);
    rtxspecial::finish(ctx->special_job);  // (ends a compilation still running for the scene replaced)
    ctx->special_job.reset();
    ctx->special_note.clear();
    ctx->info.scan_specialized = 0;
    ctx->info.specialize_seconds = 0.0f;
    if (light_tree != lights_need_tree(s)) return fail(RTX_ERR_HIP, "internal: light tree decision differs");
    const bool candidate = !use_bvh;
    if (candidate && ctx->specialize_scan) {
        if (scan_recs.size() - 1 > rtxspecial::kMaxRecords) {
            ctx->special_note = "more than " + std::to_string(rtxspecial::kMaxRecords) + " scan records";
        } else if (ctx->specialize_scan == 2u) {
            ctx->special_job = rtxspecial::start(special_text(s, light_tree, scan_groups, scan_recs, scan_info), scan_variant(s, light_tree));
            ctx->special_note = "compiling in the background";
        } else {
            double seconds = 0.0;
            if (rtxspecial::load(special_text(s, light_tree, scan_groups, scan_recs, scan_info), scan_variant(s, light_tree), ctx->special, ctx->special_note, &seconds)) {
                ctx->info.scan_specialized = 1;
                ctx->info.specialize_seconds = (float)seconds;
            }
        }
    } else if (candidate) {
        ctx->special_note = "switched off (RTX_OPTION_SPECIALIZE_SCAN)";
    }
    ctx->scene_hash = hash_scene(s, use_bvh);
    ctx->has_scene = true;
    return RTX_OK;
}

int rtx_scene_info(RtxContext* ctx, RtxSceneInfo* out) {
    if (!ctx || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->has_scene) return fail(RTX_ERR_NO_SCENE, "no scene uploaded");
    if (ctx->special_job) {  // (a finished background compilation is taken over here too)
        RTX_HIP(hipSetDevice(ctx->device));
        adopt_special_job(ctx);
    }
    *out = ctx->info;
    return RTX_OK;
}

// rtx_render_device; `may_load` = 0 for the caller that must not allocate (loading a code object does)
static int render_device(RtxContext* ctx, const RtxParams* params, float* d_accum_rgb, unsigned long long* d_rays,
                         void* hip_stream, bool may_load) {
    int rc = check_params(ctx, params);
    if (rc != RTX_OK) return rc;
    if (!d_accum_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "accumulator is null");
    RTX_HIP(hipSetDevice(ctx->device));
    if (may_load) adopt_special_job(ctx);
    rtxd::TraceArgs a = make_trace_args(ctx, params);
    a.accum = d_accum_rgb;
    a.rays = d_rays;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (params->mode == RTX_MODE_DETERMINISTIC) {
        RTX_HIP(rtx_det_mode::launch_trace(a, stream));
    } else if (ctx->special.frames) {
        RTX_HIP(launch_special_trace(ctx->special.frames, a, stream));
    } else {
        RTX_HIP(rtx_fast_mode::launch_trace(a, stream));
    }
    return RTX_OK;
}

int rtx_render_device(RtxContext* ctx, const RtxParams* params, float* d_accum_rgb,
                      unsigned long long* d_rays, void* hip_stream) {
    return render_device(ctx, params, d_accum_rgb, d_rays, hip_stream, true);
}

int rtx_render(RtxContext* ctx, const RtxParams* params, float* accum_rgb, uint64_t* rays_out) {
    int rc = check_params(ctx, params);
    if (rc != RTX_OK) return rc;
    if (!accum_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "accumulator is null");
    RTX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)params->width * params->height * 3;
    RTX_HIP(ctx->accum.resize(n));
    RTX_HIP(ctx->rays.resize(1));
    if (params->accumulate)
        RTX_HIP(hipMemcpy(ctx->accum.ptr, accum_rgb, n * sizeof(float), hipMemcpyHostToDevice));
    RTX_HIP(hipMemset(ctx->rays.ptr, 0, sizeof(unsigned long long)));
    rc = rtx_render_device(ctx, params, ctx->accum.ptr, ctx->rays.ptr, nullptr);
    if (rc != RTX_OK) return rc;
    RTX_HIP(hipDeviceSynchronize());
    RTX_HIP(hipMemcpy(accum_rgb, ctx->accum.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
    if (rays_out) {
        unsigned long long r = 0;
        RTX_HIP(hipMemcpy(&r, ctx->rays.ptr, sizeof r, hipMemcpyDeviceToHost));
        *rays_out = r;
    }
    return RTX_OK;
}

int rtx_render_noalloc(RtxContext* ctx, const RtxParams* params, float* accum_rgb, uint64_t* rays_out) {
    int rc = check_params(ctx, params);
    if (rc != RTX_OK) return rc;
    if (!accum_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "accumulator is null");
    const uint64_t n_pixels = (uint64_t)params->width * params->height;
    if (n_pixels > RTX_NOALLOC_MAX_PIXELS)
        return fail(RTX_ERR_TOO_LARGE, "rtx_render_noalloc: the frame has " + std::to_string(n_pixels) +
                                           " pixels, the static device buffer holds " +
                                           std::to_string(RTX_NOALLOC_MAX_PIXELS) + " (use rtx_render)");
    RTX_HIP(hipSetDevice(ctx->device));
    unsigned char* base = static_cast<unsigned char*>(ctx->scratch);
    unsigned long long* d_rays = reinterpret_cast<unsigned long long*>(base);
    float* d_accum = reinterpret_cast<float*>(base + kScratchCounterBytes);
    const size_t bytes = (size_t)n_pixels * 3u * sizeof(float);
    // the array keeps what the last call on this GPU left: the counter starts at zero, the frame
    // is either overwritten by the kernel (accumulate == 0) or replaced by the caller's
    RTX_HIP(hipMemsetAsync(d_rays, 0, sizeof(unsigned long long), nullptr));
    if (params->accumulate) RTX_HIP(hipMemcpy(d_accum, accum_rgb, bytes, hipMemcpyHostToDevice));
    rc = render_device(ctx, params, d_accum, d_rays, nullptr, false);
    if (rc != RTX_OK) return rc;
    RTX_HIP(hipStreamSynchronize(nullptr));
    RTX_HIP(hipMemcpy(accum_rgb, d_accum, bytes, hipMemcpyDeviceToHost));
    if (rays_out) {
        unsigned long long r = 0;
        RTX_HIP(hipMemcpy(&r, d_rays, sizeof r, hipMemcpyDeviceToHost));
        *rays_out = r;
    }
    return RTX_OK;
}

namespace {
struct EventList {  // events of one rtx_render_multi call, destroyed on every way out
    std::vector<std::pair<int, hipEvent_t>> events;
    ~EventList() {
        for (auto& e : events) {
            (void)hipSetDevice(e.first);
            (void)hipEventDestroy(e.second);
        }
    }
};
}  // namespace

int rtx_render_multi(RtxContext* const* ctxs, uint32_t n_ctx, const RtxParams* params, float* accum_rgb,
                     uint64_t* rays_out) {
    if (!ctxs || n_ctx == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "no contexts");
    if (n_ctx > 64) return fail(RTX_ERR_TOO_LARGE, "more than 64 contexts");
    if (!accum_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "accumulator is null");
    for (uint32_t i = 0; i < n_ctx; ++i) {
        if (!ctxs[i]) return fail(RTX_ERR_INVALID_ARGUMENT, "a context is null");
        for (uint32_t k = 0; k < i; ++k)
            if (ctxs[k] == ctxs[i]) return fail(RTX_ERR_INVALID_ARGUMENT, "the same context is listed twice");
        int rc = check_params(ctxs[i], params);
        if (rc != RTX_OK) return rc;
        if (ctxs[i]->scene_hash != ctxs[0]->scene_hash)
            return fail(RTX_ERR_INVALID_ARGUMENT, "the contexts do not hold the same scene");
    }
    RtxContext* root = ctxs[0];
    const size_t n = (size_t)params->width * params->height * 3;
    const size_t bytes = n * sizeof(float);

    // set-up on every device before anything is launched: buffers, counters, the caller's frame
    for (uint32_t i = 0; i < n_ctx; ++i) {
        RTX_HIP(hipSetDevice(ctxs[i]->device));
        RTX_HIP(ctxs[i]->accum.resize(n));
        RTX_HIP(ctxs[i]->rays.resize(1));
        RTX_HIP(hipMemset(ctxs[i]->rays.ptr, 0, sizeof(unsigned long long)));
        if (ctxs[i]->device != root->device) {
            int can = 0;  // direct xGMI / PCIe peer writes where the platform has them; else the copy is staged by the runtime
            if (hipDeviceCanAccessPeer(&can, ctxs[i]->device, root->device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(root->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return fail(RTX_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            }
            (void)hipGetLastError();  // "already enabled" / "no peer access" leave a sticky code behind
        }
    }
    RTX_HIP(hipSetDevice(root->device));
    if (n_ctx > 1) RTX_HIP(root->peer_frames.resize((size_t)(n_ctx - 1) * n));
    if (params->accumulate) RTX_HIP(hipMemcpy(root->accum.ptr, accum_rgb, bytes, hipMemcpyHostToDevice));

    // every device renders its block of the sample range; the launches do not wait for each other
    const uint32_t base = params->spp_count / n_ctx, longer = params->spp_count % n_ctx;
    uint32_t begin = params->spp_begin;
    for (uint32_t i = 0; i < n_ctx; ++i) {
        RtxParams p = *params;
        p.spp_begin = begin;
        p.spp_count = base + (i < longer ? 1u : 0u);
        p.accumulate = i == 0 ? params->accumulate : 0u;
        begin += p.spp_count;
        if (p.spp_count == 0 && i != 0) {  // fewer samples than contexts: an empty partial frame
            RTX_HIP(hipSetDevice(ctxs[i]->device));
            RTX_HIP(hipMemsetAsync(ctxs[i]->accum.ptr, 0, bytes, nullptr));
            continue;
        }
        int rc = rtx_render_device(ctxs[i], &p, ctxs[i]->accum.ptr, ctxs[i]->rays.ptr, nullptr);
        if (rc != RTX_OK) return rc;
    }

    // partial frames to context 0's GPU (each over its own link, behind its own render), added
    // there in context order
    EventList done;
    for (uint32_t i = 1; i < n_ctx; ++i) {
        float* slot = root->peer_frames.ptr + (size_t)(i - 1) * n;
        RTX_HIP(hipSetDevice(ctxs[i]->device));
        RTX_HIP(hipMemcpyPeerAsync(slot, root->device, ctxs[i]->accum.ptr, ctxs[i]->device, bytes, nullptr));
        hipEvent_t ev;
        RTX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        done.events.emplace_back(ctxs[i]->device, ev);
        RTX_HIP(hipEventRecord(ev, nullptr));
    }
    RTX_HIP(hipSetDevice(root->device));
    for (uint32_t i = 1; i < n_ctx; ++i) {
        RTX_HIP(hipStreamWaitEvent(nullptr, done.events[i - 1].second, 0));
        RTX_HIP(rtx_post::launch_add(root->accum.ptr, root->peer_frames.ptr + (size_t)(i - 1) * n, n, nullptr));
    }
    RTX_HIP(hipDeviceSynchronize());
    RTX_HIP(hipMemcpy(accum_rgb, root->accum.ptr, bytes, hipMemcpyDeviceToHost));
    unsigned long long total = 0;
    for (uint32_t i = 0; i < n_ctx; ++i) {
        RTX_HIP(hipSetDevice(ctxs[i]->device));
        RTX_HIP(hipDeviceSynchronize());
        unsigned long long r = 0;
        RTX_HIP(hipMemcpy(&r, ctxs[i]->rays.ptr, sizeof r, hipMemcpyDeviceToHost));
        total += r;
    }
    if (rays_out) *rays_out = total;
    return RTX_OK;
}

int rtx_render_samples(RtxContext* ctx, const RtxParams* params, float* sample_rgb,
                       uint32_t* sample_segments) {
    int rc = check_params(ctx, params);
    if (rc != RTX_OK) return rc;
    if (!sample_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "sample_rgb is null");
    const uint64_t n_samples = (uint64_t)params->width * params->height * params->spp_count;
    if (n_samples > (1ull << 28)) return fail(RTX_ERR_TOO_LARGE, "per-sample output limited to 2^28 samples");
    if (n_samples == 0) return RTX_OK;
    RTX_HIP(hipSetDevice(ctx->device));
    adopt_special_job(ctx);
    RTX_HIP(ctx->sample_rgb.resize(n_samples * 3));
    RTX_HIP(ctx->sample_segments.resize(n_samples));
    rtxd::TraceArgs a = make_trace_args(ctx, params);
    a.sample_rgb = ctx->sample_rgb.ptr;
    a.sample_segments = ctx->sample_segments.ptr;
    if (params->mode == RTX_MODE_DETERMINISTIC) RTX_HIP(rtx_det_mode::launch_trace(a, nullptr));
    else if (ctx->special.samples) RTX_HIP(launch_special_trace(ctx->special.samples, a, nullptr));
    else RTX_HIP(rtx_fast_mode::launch_trace(a, nullptr));
    RTX_HIP(hipDeviceSynchronize());
    RTX_HIP(hipMemcpy(sample_rgb, ctx->sample_rgb.ptr, n_samples * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (sample_segments)
        RTX_HIP(hipMemcpy(sample_segments, ctx->sample_segments.ptr, n_samples * sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_hit(RtxContext* ctx, uint32_t n, const float* origins, const float* directions, float t0,
            float t1, uint32_t mode, RtxHit* hits) {
    if (!ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "context is null");
    if (!ctx->has_scene) return fail(RTX_ERR_NO_SCENE, "no scene uploaded");
    if (mode > RTX_MODE_DETERMINISTIC) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown mode");
    if (n == 0) return RTX_OK;
    if (!origins || !directions || !hits) return fail(RTX_ERR_INVALID_ARGUMENT, "null ray / hit array");
    if (n > (1u << 28)) return fail(RTX_ERR_TOO_LARGE, "more than 2^28 rays");
    RTX_HIP(hipSetDevice(ctx->device));
    adopt_special_job(ctx);
    RTX_HIP(ctx->ray_o.resize((size_t)n * 3));
    RTX_HIP(ctx->ray_d.resize((size_t)n * 3));
    RTX_HIP(ctx->hits.resize(n));
    RTX_HIP(hipMemcpy(ctx->ray_o.ptr, origins, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(ctx->ray_d.ptr, directions, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
    rtxd::HitArgs a{};
    a.scene = ctx->view;
    a.n = n;
    a.t0 = t0;
    a.t1 = t1;
    a.origins = ctx->ray_o.ptr;
    a.directions = ctx->ray_d.ptr;
    a.hits = ctx->hits.ptr;
    if (mode == RTX_MODE_DETERMINISTIC) RTX_HIP(rtx_det_mode::launch_hit(a, nullptr));
    else if (ctx->special.hit) RTX_HIP(launch_special_hit(ctx->special.hit, a, nullptr));
    else RTX_HIP(rtx_fast_mode::launch_hit(a, nullptr));
    RTX_HIP(hipDeviceSynchronize());
    RTX_HIP(hipMemcpy(hits, ctx->hits.ptr, (size_t)n * sizeof(RtxHit), hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_resolve_device(RtxContext* ctx, uint32_t width, uint32_t height, uint32_t ns,
                       const float* d_accum_rgb, uint8_t* rgb8, void* hip_stream) {
    if (!ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "context is null");
    if (!d_accum_rgb || !rgb8) return fail(RTX_ERR_INVALID_ARGUMENT, "null buffer");
    if (width == 0 || height == 0 || ns == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "empty image or ns == 0");
    if ((uint64_t)width * height > (1ull << 30)) return fail(RTX_ERR_TOO_LARGE, "more than 2^30 pixels");
    RTX_HIP(hipSetDevice(ctx->device));
    const uint32_t n_pixels = width * height;
    RTX_HIP(ctx->rgb8.resize((size_t)n_pixels * 3));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    RTX_HIP(rtx_post::launch_resolve(d_accum_rgb, ctx->rgb8.ptr, n_pixels, ns, stream));
    RTX_HIP(hipMemcpyAsync(rgb8, ctx->rgb8.ptr, (size_t)n_pixels * 3, hipMemcpyDeviceToHost, stream));
    RTX_HIP(hipStreamSynchronize(stream));
    return RTX_OK;
}

int rtx_resolve_to_device(RtxContext* ctx, uint32_t width, uint32_t height, uint32_t ns,
                          const float* d_accum_rgb, uint8_t* d_rgb8, void* hip_stream) {
    if (!ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "context is null");
    if (!d_accum_rgb || !d_rgb8) return fail(RTX_ERR_INVALID_ARGUMENT, "null buffer");
    if (width == 0 || height == 0 || ns == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "empty image or ns == 0");
    if ((uint64_t)width * height > (1ull << 30)) return fail(RTX_ERR_TOO_LARGE, "more than 2^30 pixels");
    RTX_HIP(hipSetDevice(ctx->device));
    RTX_HIP(rtx_post::launch_resolve(d_accum_rgb, d_rgb8, width * height, ns, static_cast<hipStream_t>(hip_stream)));
    return RTX_OK;
}

}  // extern "C"
