// scene_factory.cpp — named benchmark / test scenes built through the host scene API,
// and a small C interface so that Python tests and bench.py can obtain the
// flattened RtxScene of a named scene and exercise the host-side image pipeline.
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "raytrace.h"

using namespace rt;

// internal linkage: the oracle build includes the same file against other classes
namespace {
#include "scenes.inc"
}  // namespace

namespace {
thread_local std::string g_error;

struct NamedScene : public Scene {
    NamedScene(const std::string& name, const char* file, int w, int h, int spp)
        : Scene(file, w, h, spp), m_name(name) {}
    void build() override {
        if (m_name == "cornell_builtin") {
            Scene::build();
            m_built = true;
            return;
        }
        SceneParts parts;
        m_built = make_scene(m_name.c_str(), m_image->width(), m_image->height(), parts);
        if (!m_built) return;
        setCamera(parts.camera);
        setWorld(parts.world);
        setLight(parts.light);
        setBackground(parts.background);
        setBackgroundSky(parts.sky);
        setGeneralLightSampling(parts.general_lights);
    }
    bool built() const { return m_built; }

private:
    std::string m_name;
    bool m_built = false;
};

struct Handle {
    FlatScene flat;
};
}  // namespace

extern "C" {

const char* rtxh_last_error(void) { return g_error.c_str(); }

// Build a named scene ("cornell_builtin" = Scene::build(); otherwise scenes.inc) and
// flatten it.  Returns an opaque handle or NULL.
void* rtxh_scene_create(const char* name, int w, int h) {
    NamedScene scene(name, "unused.bmp", w, h, 1);
    scene.build();
    if (!scene.built()) {
        g_error = std::string("unknown scene '") + name + "'";
        return nullptr;
    }
    Handle* hd = new Handle();
    if (!scene.flatten(hd->flat)) {
        g_error = hd->flat.error();
        delete hd;
        return nullptr;
    }
    return hd;
}

const RtxScene* rtxh_scene_view(void* handle) {
    return handle ? static_cast<Handle*>(handle)->flat.view() : nullptr;
}

void rtxh_scene_destroy(void* handle) { delete static_cast<Handle*>(handle); }

// Image::write for n pixels laid out in a row.
void rtxh_image_write(int n, const float* rgb, uint8_t* out) {
    Image img(n, 1);
    for (int k = 0; k < n; ++k) img.write(k, 0, rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]);
    std::memcpy(out, img.pixels(), (size_t)n * 3);
}

// sum / ns then Image::write for a whole frame (rows top-first).
void rtxh_resolve(int w, int h, int ns, const float* sum_rgb, uint8_t* out) {
    Image img(w, h);
    const float fns = (float)ns;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* c = sum_rgb + ((size_t)y * w + x) * 3;
            img.write(x, y, c[0] / fns, c[1] / fns, c[2] / fns);
        }
    std::memcpy(out, img.pixels(), (size_t)w * h * 3);
}

int rtxh_write_bmp(const char* path, int w, int h, const uint8_t* rgb) {
    Image img(w, h);
    std::memcpy(img.pixels(), rgb, (size_t)w * h * 3);
    return img.save_bmp(path) ? 1 : 0;
}

// The texels rt::ImageTexture holds for a file (24-bit BMP, binary PPM, sequential JPEG): RGB,
// rows top-down.  Returns 1 and the size (and the texels, if `out` holds w * h * 3 bytes), 0 and
// rtxh_last_error() if the file cannot be decoded.
int rtxh_load_image(const char* path, int* w, int* h, uint8_t* out, size_t capacity) {
    ImageTexture tex(path);
    if (!tex.loaded()) {
        g_error = std::string("cannot decode '") + (path ? path : "") + "'" + (tex.error().empty() ? "" : ": " + tex.error());
        return 0;
    }
    *w = tex.width();
    *h = tex.height();
    const size_t bytes = (size_t)tex.width() * tex.height() * 3;
    if (out && bytes <= capacity) std::memcpy(out, tex.texels(), bytes);
    return 1;
}

// Scene::setKernelCompilation for the renders below (-1 auto, 0 never, 1 at upload, 2 in the background).
static int g_kernel_compilation = -1;
int rtxh_set_kernel_compilation(int mode) {
    if (mode < -1 || mode > 2) {
        g_error = "kernel compilation mode must be -1 (auto), 0 (never), 1 (at upload) or 2 (in the background)";
        return 0;
    }
    g_kernel_compilation = mode;
    return 1;
}

// The whole Scene::render() path: build, flatten, GPU render, filters, BMP.
int rtxh_render_to_bmp(const char* name, const char* path, int w, int h, int spp, int deterministic,
                       uint64_t seed, int device, unsigned long long* rays, double* seconds) {
    NamedScene scene(name, path, w, h, spp);
    scene.setDeterministic(deterministic != 0);
    scene.setSeed(seed);
    scene.setDevice(device);
    scene.setKernelCompilation(static_cast<Scene::KernelCompilation>(g_kernel_compilation));
    if (!scene.render()) {
        g_error = scene.built() ? scene.lastError() : std::string("unknown scene '") + name + "'";
        return 0;
    }
    if (rays) *rays = scene.raysTraced();
    if (seconds) *seconds = scene.renderSeconds();
    return 1;
}

// The same on several GPUs of this host (Scene::setDevices).
int rtxh_render_to_bmp_multi(const char* name, const char* path, int w, int h, int spp, int deterministic,
                             uint64_t seed, const int* devices, int n_devices, unsigned long long* rays,
                             double* seconds) {
    if (!devices || n_devices <= 0) {
        g_error = "no devices";
        return 0;
    }
    NamedScene scene(name, path, w, h, spp);
    scene.setDeterministic(deterministic != 0);
    scene.setSeed(seed);
    scene.setDevices(std::vector<int>(devices, devices + n_devices));
    scene.setKernelCompilation(static_cast<Scene::KernelCompilation>(g_kernel_compilation));
    if (!scene.render()) {
        g_error = scene.built() ? scene.lastError() : std::string("unknown scene '") + name + "'";
        return 0;
    }
    if (rays) *rays = scene.raysTraced();
    if (seconds) *seconds = scene.renderSeconds();
    return 1;
}

}  // extern "C"
