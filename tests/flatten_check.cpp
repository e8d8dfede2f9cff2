// flatten_check.cpp — host-only checker of the scene flattener (raytrace_amd/host/raytrace.cpp):
// builds every named scene through the host scene API, flattens it and reads every element of
// every array the RtxScene names, checking the indices that link them.  Test infrastructure:
// compiled by tests/test_host_sanitizers.py with AddressSanitizer and UndefinedBehaviorSanitizer
// (an out-of-bounds count or a dangling array shows as a sanitizer report), no GPU involved.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rtx.h"

extern "C" {
void* rtxh_scene_create(const char* name, int w, int h);
const RtxScene* rtxh_scene_view(void* handle);
void rtxh_scene_destroy(void* handle);
const char* rtxh_last_error(void);
void rtxh_image_write(int n, const float* rgb, uint8_t* out);
int rtxh_write_bmp(const char* path, int w, int h, const uint8_t* rgb);
}

static int fail(const std::string& scene, const char* what) {
    std::printf("FAILED %s: %s\n", scene.c_str(), what);
    return 1;
}

template <typename T>
static double sum(const T* p, size_t n) {
    double s = 0.0;
    for (size_t k = 0; k < n; ++k) s += (double)p[k];
    return s;
}

int main(int argc, char** argv) {
    const char* names[] = {"cornell_builtin", "cornell", "cornell_classic", "cornell_extra", "small", "random_field",
                           "bvh_stress_300", "sky_spheres", "sky_sphere_lit", "cornell_light_tree", "cornell_many_lights",
                           "cornell_many_lights_10", "cornell_side_lights", "cornell_side_lights_nolight", "textured"};
    double checksum = 0.0;
    for (const char* name : names) {
        void* h = rtxh_scene_create(name, 37, 21);
        if (!h) {
            if (std::string(name) == "textured") continue;  // needs the image file of the test suite
            return fail(name, rtxh_last_error());
        }
        const RtxScene* s = rtxh_scene_view(h);
        if (!s || s->abi_version != RTX_ABI_VERSION) return fail(name, "no view / ABI version");
        if (s->n_prims == 0 || s->n_chains == 0 || s->n_materials == 0) return fail(name, "empty tables");
        for (uint32_t p = 0; p < s->n_prims; ++p) {
            if (s->prim_kind[p] > RTX_PRIM_SPHERE || s->prim_flip[p] > 1u) return fail(name, "prim kind / flip");
            if (s->prim_material[p] >= s->n_materials) return fail(name, "prim material");
            if (s->prim_chain[p] >= s->n_chains) return fail(name, "prim chain");
        }
        checksum += sum(s->prim_a, s->n_prims) + sum(s->prim_b, s->n_prims) + sum(s->prim_c, s->n_prims) +
                    sum(s->prim_d, s->n_prims) + sum(s->prim_e, s->n_prims);
        if (s->chain_begin[0] != 0 || s->chain_begin[s->n_chains] != s->n_steps) return fail(name, "chain table");
        for (uint32_t c = 0; c < s->n_chains; ++c)
            if (s->chain_begin[c] > s->chain_begin[c + 1]) return fail(name, "chain order");
        for (uint32_t k = 0; k < s->n_steps; ++k) {
            if (s->step_kind[k] > RTX_STEP_ROTATE) return fail(name, "step kind");
            if (s->step_kind[k] == RTX_STEP_ROTATE) {
                const double q = (double)s->step_x[k] * s->step_x[k] + (double)s->step_y[k] * s->step_y[k] +
                                 (double)s->step_z[k] * s->step_z[k] + (double)s->step_w[k] * s->step_w[k];
                if (std::fabs(q - 1.0) > 1e-5) return fail(name, "rotation is not a unit quaternion");
            }
        }
        for (uint32_t m = 0; m < s->n_materials; ++m) {
            if (s->mat_kind[m] > RTX_MAT_DIFFUSE_LIGHT) return fail(name, "material kind");
            if (s->mat_kind[m] != RTX_MAT_DIELECTRIC && s->mat_texture[m] >= s->n_textures) return fail(name, "material texture");
            checksum += s->mat_param[m];
        }
        for (uint32_t t = 0; t < s->n_textures; ++t) {
            if (s->tex_kind[t] > RTX_TEX_SKY) return fail(name, "texture kind");
            if (s->tex_kind[t] == RTX_TEX_CHECKER && (s->tex_odd[t] >= s->n_textures || s->tex_even[t] >= s->n_textures))
                return fail(name, "checker sub-texture");
            if (s->tex_kind[t] == RTX_TEX_IMAGE &&
                (uint64_t)s->tex_offset[t] + 3ull * s->tex_odd[t] * s->tex_even[t] > s->n_texel_bytes)
                return fail(name, "image outside the texel buffer");
            checksum += s->tex_r[t] + s->tex_g[t] + s->tex_b[t];
        }
        checksum += sum(s->texels, (size_t)s->n_texel_bytes);
        const uint32_t n_le = s->n_light_entries ? s->n_light_entries : s->n_lights;
        if (n_le < s->n_lights) return fail(name, "light entries");
        for (uint32_t l = 0; l < n_le; ++l) {
            if (s->light_kind[l] > RTX_LIGHT_LIST) return fail(name, "light kind");
            checksum += s->light_a[l] + s->light_b[l] + s->light_c[l] + s->light_d[l] + s->light_e[l];
            if (s->light_kind[l] == RTX_LIGHT_LIST) {
                if (!s->light_first || !s->light_count) return fail(name, "nested list without ranges");
                if (s->light_first[l] <= l || s->light_first[l] + s->light_count[l] > n_le || s->light_count[l] == 0)
                    return fail(name, "nested list range");
            } else if (s->light_chain && s->light_chain[l] >= s->n_chains) {
                return fail(name, "light chain");
            }
        }
        for (int c = 0; c < 3; ++c) checksum += s->cam_origin[c] + s->cam_ll[c] + s->cam_u[c] + s->cam_v[c] + s->background[c];
        rtxh_scene_destroy(h);
    }
    if (rtxh_scene_create("no_such_scene", 8, 8)) return fail("no_such_scene", "an unknown name made a scene");
    // the image pipeline: filter chain incl. NaN / inf, BMP writer with row padding
    const float px[] = {0.5f, NAN, -1.0f, INFINITY, 0.25f, 1e30f, 0.0f, -0.0f, 2.0f};
    uint8_t out[9];
    rtxh_image_write(3, px, out);
    const std::string bmp = argc > 1 ? argv[1] : "/tmp/flatten_check.bmp";
    std::vector<uint8_t> img(7 * 5 * 3, 77);
    if (!rtxh_write_bmp(bmp.c_str(), 7, 5, img.data())) return fail("bmp", "write failed");
    std::printf("ok %zu scenes, checksum %.6e, bytes %u %u %u\n", sizeof names / sizeof names[0], checksum, out[0], out[1], out[2]);
    return 0;
}
