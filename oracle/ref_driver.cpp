/*
 * ref_driver.cpp — C interface around the UNMODIFIED reference tracer, built
 * into oracle/_ref/libref_oracle.so by oracle/Makefile together with the
 * reference's own translation units (compiled in place from /root/reference).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, by __graft_entry__.smoke() and by
 * the cpu_baseline leg of bench.py as the checker / the reported CPU baseline.
 * The renderer never loads this library.
 *
 * What it drives is the reference's public integrator
 *   Vector3 Scene::color(const Ray&, const Shape* world, const Shape* light, int depth)
 * (Raytrace_C++/Src/Scene.cpp:103-132) with the pixel / sample loop of
 * Scene::render() (Scene.cpp:140-151) restated here because the reference's
 * own loop writes straight into an 8-bit image and cannot expose per-sample
 * radiance.  The random stream is the counter-based one of include/rtx_det.h,
 * substituted for rand() by oracle/ref_shim.h.
 */
#define private public /* reach Scene::m_world / m_light / m_camera / m_backColor (Scene.h:37-43) and, for
                          the general light shapes below, what the shape classes hold (Rect.h:31-33,
                          Translate.h:15-16, Rotate.h:15-16, FlipNormals.h:14, Box.h:14-15, ShapeList.h:19) */
#include "Scene.h"
#include "ShapeList.h"
#include "ShapeBuilder.h"
#undef private

#include "HitRec.h"
#include "ScatterRec.h"
#include "Lambertian.h"
#include "Metal.h"
#include "Dielectric.h"
#include "DiffuseLight.h"
#include "ColorTexture.h"
#include "CheckerTexture.h"
#include "CosinePdf.h"
#include <stb_image.h> /* declarations only, for ImageTexture.h:9 */
#include "ImageTexture.h"
#include <stb_image_write.h> /* declarations only; the implementation is compiled in Scene.cpp:3-6 */

// A texture the reference does not have, written against its Texture interface (Texture.h:5):
// value(u, v, p) = the reference's own Scene::backgroundSky(p - centre) (Scene.h:24-28).  On an
// emitting sphere around the scene it lets the UNMODIFIED Scene::color render - and sample as a
// light - the sky gradient it never calls itself (scene "sky_sphere_lit").
class SkyTexture : public Texture {
public:
    SkyTexture(const Vector3& centre) : m_centre(centre), m_scene("unused.bmp", 1, 1, 1) {}
    virtual Vector3 value(float u, float v, const Vector3& p) const override { return m_scene.backgroundSky(p - m_centre); }

private:
    Vector3 m_centre;
    Scene m_scene;
};

// General light shapes.  In the reference only an XZ rect and a sphere can be sampled as lights:
// Rect::pdf_value / random give up on the other two axes (Rect.cpp:37,51) and FlipNormals, Translate,
// Rotate and Box keep Shape's defaults, 0 and (1,0,0) (Shape.h:8-9).  The renderer's opt-in general
// light sampling (RTX_LIGHTS_GENERAL) gives all of them a density.  These classes state the same
// extension against the reference's Shape interface, so that a light shape built from them runs
// through the UNMODIFIED Scene::color -> MixturePdf -> ShapePdf -> ShapeList::pdf_value / random
// (Scene.cpp:112-120, MixturePdf.h:7-22, ShapePdf.h:8-17, ShapeList.cpp:19-35): the mixture, the
// list weights, the draw order and the integrator are the reference's, only the leaf densities
// and the way a query passes a wrapper are the extension's.
class GeneralRect : public Shape {  // a rect of any axis, sampled by area like the XZ one
public:
    explicit GeneralRect(const Rect& r) : m_rect(r) {}
    bool hit(const Ray& r, float t0, float t1, HitRec& hrec) const override { return m_rect.hit(r, t0, t1, hrec); }
    float pdf_value(const Vector3& o, const Vector3& v) const override {  // Rect.cpp:38-47 for the rect's own axis
        HitRec hrec;
        if (!m_rect.hit(Ray(o, v), 0.001f, FLT_MAX, hrec)) return 0;
        const float area = (m_rect.m_x1 - m_rect.m_x0) * (m_rect.m_y1 - m_rect.m_y0);
        const float distance_squared = pow2(hrec.t) * lengthSqr(v);
        const float cosine = fabs(dot(v, hrec.n)) / length(v);
        return distance_squared / (cosine * area);
    }
    Vector3 random(const Vector3& o) const override {  // Rect.cpp:52-67 for the rect's own axis
        const float x = m_rect.m_x0 + drand48() * (m_rect.m_x1 - m_rect.m_x0);
        const float y = m_rect.m_y0 + drand48() * (m_rect.m_y1 - m_rect.m_y0);
        const float k = m_rect.m_k;
        const Vector3 point = m_rect.m_axis == Rect::kXY ? Vector3(x, y, k)
                            : m_rect.m_axis == Rect::kXZ ? Vector3(x, k, y) : Vector3(k, x, y);
        return point - o;
    }

private:
    Rect m_rect;
};
class LightFlip : public Shape {  // FlipNormals: the densities use |cos|, a flip changes nothing
public:
    explicit LightFlip(const ShapePtr& s) : m_shape(s) {}
    bool hit(const Ray& r, float t0, float t1, HitRec& hrec) const override { return m_shape->hit(r, t0, t1, hrec); }
    float pdf_value(const Vector3& o, const Vector3& v) const override { return m_shape->pdf_value(o, v); }
    Vector3 random(const Vector3& o) const override { return m_shape->random(o); }

private:
    ShapePtr m_shape;
};
class LightTranslate : public Shape {  // the query enters the frame as a ray enters Translate::hit (Translate.cpp:7)
public:
    LightTranslate(const ShapePtr& s, const Vector3& offset) : m_shape(s), m_offset(offset) {}
    bool hit(const Ray& r, float t0, float t1, HitRec& hrec) const override { return false; }
    float pdf_value(const Vector3& o, const Vector3& v) const override { return m_shape->pdf_value(o - m_offset, v); }
    Vector3 random(const Vector3& o) const override { return m_shape->random(o - m_offset); }

private:
    ShapePtr m_shape;
    Vector3 m_offset;
};
class LightRotate : public Shape {  // ... and Rotate::hit (Rotate.cpp:7-9); a direction leaves as a normal does (Rotate.cpp:13)
public:
    LightRotate(const ShapePtr& s, const Quat& q) : m_shape(s), m_quat(q) {}
    bool hit(const Ray& r, float t0, float t1, HitRec& hrec) const override { return false; }
    float pdf_value(const Vector3& o, const Vector3& v) const override {
        const Quat revq = conj(m_quat);
        return m_shape->pdf_value(rotate(revq, o), rotate(revq, v));
    }
    Vector3 random(const Vector3& o) const override { return rotate(m_quat, m_shape->random(rotate(conj(m_quat), o))); }

private:
    ShapePtr m_shape;
    Quat m_quat;
};
// The general counterpart of a light shape built through the reference's ShapeBuilder: a Box
// becomes the list of its faces (the reference's own ShapeList, so one more draw picks the face
// and each face weighs 1/6), a sphere stays the reference's Sphere.
static ShapePtr general_light(const ShapePtr& s) {
    if (const Rect* r = dynamic_cast<const Rect*>(s.get())) return std::make_shared<GeneralRect>(*r);
    if (const FlipNormals* f = dynamic_cast<const FlipNormals*>(s.get())) return std::make_shared<LightFlip>(general_light(f->m_shape));
    if (const Translate* t = dynamic_cast<const Translate*>(s.get()))
        return std::make_shared<LightTranslate>(general_light(t->m_shape), t->m_offset);
    if (const Rotate* r = dynamic_cast<const Rotate*>(s.get())) return std::make_shared<LightRotate>(general_light(r->m_shape), r->m_quat);
    const ShapeList* list = dynamic_cast<const ShapeList*>(s.get());
    if (const Box* b = dynamic_cast<const Box*>(s.get())) list = b->m_list.get();
    if (list) {
        auto out = std::make_shared<ShapeList>();
        for (const ShapePtr& c : list->m_list) out->add(general_light(c));
        return out;
    }
    return s;
}

// internal linkage: the renderer's host library includes the same file against its own classes
namespace {
#include "../raytrace_amd/host/scenes.inc"
}  // namespace

thread_local RtxRng g_rt_rng = {0u, 1u};
thread_local unsigned long long g_rt_draws = 0;
static thread_local unsigned long long tl_segments = 0;

namespace {

// Counts nearest-hit queries issued by Scene::color (one per path segment).
class CountingWorld : public Shape {
public:
    explicit CountingWorld(const Shape* inner) : m_inner(inner) {}
    bool hit(const Ray& r, float t0, float t1, HitRec& hrec) const override {
        ++tl_segments;
        return m_inner->hit(r, t0, t1, hrec);
    }

private:
    const Shape* m_inner;
};

struct RefScene {
    std::unique_ptr<Scene> scene;  // supplies color() and the background
    const Shape* world = nullptr;
    const Shape* light = nullptr;
    const Camera* camera = nullptr;
    std::unique_ptr<Camera> own_camera;
    std::unique_ptr<ShapeList> own_world, own_light;
    std::unique_ptr<CountingWorld> counted;
    int w = 0, h = 0;
};

int material_kind(const MaterialPtr& m) {
    if (!m) return -1;
    if (dynamic_cast<Lambertian*>(m.get())) return 0;
    if (dynamic_cast<Metal*>(m.get())) return 1;
    if (dynamic_cast<Dielectric*>(m.get())) return 2;
    if (dynamic_cast<DiffuseLight*>(m.get())) return 3;
    return 4;
}

}  // namespace

extern "C" {

/*
 * name == "cornell_builtin": world / light / camera come from the reference's own
 * Scene::build().  Any other name: built by scenes.inc through the reference's
 * ShapeBuilder API.
 */
void* ref_scene_create(const char* name, int w, int h) {
    auto rs = std::make_unique<RefScene>();
    rs->w = w;
    rs->h = h;
    rs->scene = std::make_unique<Scene>("unused.bmp", w, h, 1);
    if (std::string(name) == "cornell_builtin") {
        rs->scene->build();
        rs->world = rs->scene->m_world.get();
        rs->light = rs->scene->m_light.get();
        rs->camera = rs->scene->m_camera.get();
    } else {
        SceneParts parts;
        if (!make_scene(name, w, h, parts)) return nullptr;
        if (parts.sky || !parts.light) {  // the reference's color() cannot render these
            delete parts.camera;
            delete parts.world;
            delete parts.light;
            return nullptr;
        }
        rs->own_camera.reset(parts.camera);
        rs->own_world.reset(parts.world);
        rs->own_light.reset(parts.light);
        if (parts.general_lights) {  // the same light list made of the general shapes above
            auto general = std::make_unique<ShapeList>();
            for (const ShapePtr& c : parts.light->m_list) general->add(general_light(c));
            rs->own_light = std::move(general);
        }
        rs->scene->m_backColor = parts.background;
        rs->world = rs->own_world.get();
        rs->light = rs->own_light.get();
        rs->camera = rs->own_camera.get();
    }
    rs->counted = std::make_unique<CountingWorld>(rs->world);
    return rs.release();
}

void ref_scene_destroy(void* p) { delete static_cast<RefScene*>(p); }

/*
 * Radiance of samples [spp_begin, spp_begin+spp_count) of every pixel.
 *   sum_rgb            [h][w][3], row 0 = TOP image row (the reference stores pixel
 *                      (i, j) at row ny-j-1, Scene.cpp:151); fp32 running sum in
 *                      sample order, i.e. the reference's `c` before `c /= ns`.
 *   per_sample_rgb     optional [h*w][spp_count][3], pixel index = j*w+i, j from the
 *                      bottom (the reference's loop variables).
 *   per_sample_segments optional [h*w][spp_count]: world->hit calls of that sample.
 */
int ref_render(void* p, int spp_begin, int spp_count, uint64_t seed, int n_threads,
               float* sum_rgb, float* per_sample_rgb, uint32_t* per_sample_segments,
               uint64_t* total_rays, uint64_t* total_draws) {
    RefScene* rs = static_cast<RefScene*>(p);
    if (!rs || spp_count < 0) return -1;
    const int nx = rs->w, ny = rs->h;
    std::atomic<int> next_row(0);
    std::atomic<unsigned long long> rays(0), draws(0);
    auto worker = [&]() {
        unsigned long long my_rays = 0, my_draws = 0;
        for (;;) {
            int j = next_row.fetch_add(1);
            if (j >= ny) break;
            for (int i = 0; i < nx; ++i) {
                Vector3 c(0);
                const uint32_t pixel = uint32_t(j) * uint32_t(nx) + uint32_t(i);
                for (int s = 0; s < spp_count; ++s) {
                    rtx_rng_init(&g_rt_rng, pixel, uint32_t(spp_begin + s), seed);
                    g_rt_draws = 0;
                    tl_segments = 0;
                    float u = (float(i) + rt_reference_drand48()) / float(nx);
                    float v = (float(j) + rt_reference_drand48()) / float(ny);
                    Ray r = rs->camera->getRay(u, v);
                    Vector3 L = rs->scene->color(r, rs->counted.get(), rs->light, 0);
                    c += L;
                    my_rays += tl_segments;
                    my_draws += g_rt_draws;
                    if (per_sample_rgb) {
                        float* o = per_sample_rgb + (size_t(pixel) * spp_count + s) * 3;
                        o[0] = L.getX();
                        o[1] = L.getY();
                        o[2] = L.getZ();
                    }
                    if (per_sample_segments)
                        per_sample_segments[size_t(pixel) * spp_count + s] = uint32_t(tl_segments);
                }
                if (sum_rgb) {
                    float* o = sum_rgb + (size_t(ny - j - 1) * nx + i) * 3;
                    o[0] = c.getX();
                    o[1] = c.getY();
                    o[2] = c.getZ();
                }
            }
        }
        rays += my_rays;
        draws += my_draws;
    };
    if (n_threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
    if (total_rays) *total_rays = rays.load();
    if (total_draws) *total_draws = draws.load();
    return 0;
}

/* Nearest hit of n arbitrary rays against the scene's world (Shape::hit, Shape.h:7). */
int ref_hit(void* p, int n, const float* org, const float* dir, float t0, float t1, int32_t* hit,
            float* t, float* pos, float* nrm, float* uv, int32_t* mat_kind) {
    RefScene* rs = static_cast<RefScene*>(p);
    if (!rs) return -1;
    for (int k = 0; k < n; ++k) {
        Ray r(Vector3(org[3 * k], org[3 * k + 1], org[3 * k + 2]),
              Vector3(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]));
        HitRec hrec;
        bool h = rs->world->hit(r, t0, t1, hrec);
        hit[k] = h ? 1 : 0;
        if (h) {
            t[k] = hrec.t;
            for (int c = 0; c < 3; ++c) {
                pos[3 * k + c] = hrec.p[c];
                nrm[3 * k + c] = hrec.n[c];
            }
            uv[2 * k] = hrec.u;
            uv[2 * k + 1] = hrec.v;
            mat_kind[k] = material_kind(hrec.mat);
        } else {
            t[k] = 0.0f;
            for (int c = 0; c < 3; ++c) pos[3 * k + c] = nrm[3 * k + c] = 0.0f;
            uv[2 * k] = uv[2 * k + 1] = 0.0f;
            mat_kind[k] = -1;
        }
    }
    return 0;
}

/* Camera::getRay (Camera.h:24-26) for n (u,v) pairs. */
int ref_camera_rays(void* p, int n, const float* uv, float* org, float* dir) {
    RefScene* rs = static_cast<RefScene*>(p);
    if (!rs) return -1;
    for (int k = 0; k < n; ++k) {
        Ray r = rs->camera->getRay(uv[2 * k], uv[2 * k + 1]);
        for (int c = 0; c < 3; ++c) {
            org[3 * k + c] = r.origin()[c];
            dir[3 * k + c] = r.direction()[c];
        }
    }
    return 0;
}

/* Image::write (Image.h:30-39): NaN filter, gamma, clamp, 8-bit quantisation. */
void ref_image_write(int n, const float* rgb, uint8_t* out) {
    Image img(n, 1);
    for (int k = 0; k < n; ++k) img.write(k, 0, rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]);
    memcpy(out, img.pixels(), size_t(n) * 3);
}

/*
 * The reference's per-pixel epilogue (Scene.cpp:150-151): c /= ns, Image::write at
 * row ny-j-1.  `sum_rgb` is laid out as ref_render() returns it (row 0 = top).
 */
void ref_resolve(int w, int h, int ns, const float* sum_rgb, uint8_t* out) {
    Image img(w, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* s = sum_rgb + (size_t(y) * w + x) * 3;
            Vector3 c(s[0], s[1], s[2]);
            c /= ns;
            img.write(x, y, c.getX(), c.getY(), c.getZ());
        }
    memcpy(out, img.pixels(), size_t(w) * h * 3);
}

/* The whole reference pipeline including its BMP writer (Scene.cpp:155). */
int ref_write_bmp(const char* path, int w, int h, const uint8_t* rgb) {
    return stbi_write_bmp(path, w, h, 3, rgb);
}

/* The texels the reference's ImageTexture holds for a file: the loader call of ImageTexture.h:9
 * (stbi_load with no forced channel count).  Returns the channel count, 0 if the file cannot be
 * read; `out` (may be null: query of the size) receives w * h * channels bytes. */
int ref_load_image(const char* path, int* w, int* h, uint8_t* out, size_t capacity) {
    int n = 0;
    unsigned char* texels = stbi_load(path, w, h, &n, 0);
    if (!texels) return 0;
    const size_t bytes = size_t(*w) * size_t(*h) * size_t(n);
    if (out && bytes <= capacity) memcpy(out, texels, bytes);
    stbi_image_free(texels);
    return n;
}

/* A JPEG file of an RGB image, written by the encoder the reference vendors beside its BMP
 * writer (external/stb/stb_image_write.h): test input for the product's JPEG decoder. */
int ref_write_jpg(const char* path, int w, int h, int channels, const uint8_t* pixels, int quality) {
    return stbi_write_jpg(path, w, h, channels, pixels, quality);
}

/* Scene::backgroundSky (Scene.h:24-28) of n directions: the reference defines the gradient but
 * its color() never calls it, so it is pinned function by function. */
int ref_background_sky(int n, const float* dirs, float* out) {
    Scene scene("unused.bmp", 4, 4, 1);
    for (int k = 0; k < n; ++k) {
        Vector3 c = scene.backgroundSky(Vector3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
        out[3 * k] = c.getX();
        out[3 * k + 1] = c.getY();
        out[3 * k + 2] = c.getZ();
    }
    return 0;
}

/* CosinePdf::generate / ::value (CosinePdf.cpp:6-20) around n normals; entry k draws from the
 * stream of (pixel k, sample 0, seed). */
int ref_cosine_pdf(int n, const float* normals, uint64_t seed, float* dirs, float* values) {
    CosinePdf pdf;
    for (int k = 0; k < n; ++k) {
        HitRec h;
        h.n = Vector3(normals[3 * k], normals[3 * k + 1], normals[3 * k + 2]);
        rtx_rng_init(&g_rt_rng, uint32_t(k), 0u, seed);
        Vector3 d = pdf.generate(h);
        dirs[3 * k] = d.getX();
        dirs[3 * k + 1] = d.getY();
        dirs[3 * k + 2] = d.getZ();
        values[k] = pdf.value(h, d);
    }
    return 0;
}
/* Shape::pdf_value / Shape::random of the scene's light shape (ShapeList.cpp:19-35,
 * Rect.cpp:36-68, Sphere.cpp:38-58, Shape.h:8-9) for n (origin, direction) pairs; entry k draws
 * its random() from the stream of (pixel k, sample 0, seed). */
int ref_light_pdf(void* p, int n, const float* org, const float* dir, uint64_t seed, float* values, float* dirs) {
    RefScene* rs = static_cast<RefScene*>(p);
    if (!rs || !rs->light) return -1;
    for (int k = 0; k < n; ++k) {
        const Vector3 o(org[3 * k], org[3 * k + 1], org[3 * k + 2]);
        const Vector3 v(dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]);
        values[k] = rs->light->pdf_value(o, v);
        rtx_rng_init(&g_rt_rng, uint32_t(k), 0u, seed);
        const Vector3 d = rs->light->random(o);
        dirs[3 * k] = d.getX();
        dirs[3 * k + 1] = d.getY();
        dirs[3 * k + 2] = d.getZ();
    }
    return 0;
}

/* Material::scatter / ::emitted / ::scattering_pdf (Material.h:9-11; Lambertian.cpp:16-25, Metal.cpp:9-17,
 * Dielectric.cpp:7-45, DiffuseLight.h:17-28) of one material - kind 0 Lambertian, 1 Metal (param = fuzz),
 * 2 Dielectric (param = refractive index), 3 DiffuseLight, over ColorTexture(tex) - for n (incoming
 * direction, hit normal, scattered direction) triples at the hit point (0.5, 0.25, -1); entry k draws
 * from the stream of (pixel k, sample 0, seed).  A material that does not scatter leaves dir at 0;
 * Lambertian leaves it at 0 too (its direction is drawn by Scene::color from srec.pdf). */
int ref_scatter(uint32_t kind, float param, const float* tex, int n, const float* d_in, const float* normals,
                const float* scattered, uint64_t seed, int32_t* ok, int32_t* specular, float* dir, float* albedo,
                float* emitted, float* spdf) {
    TexturePtr t = std::make_shared<ColorTexture>(Vector3(tex[0], tex[1], tex[2]));
    MaterialPtr m;
    if (kind == 0) m = std::make_shared<Lambertian>(t);
    else if (kind == 1) m = std::make_shared<Metal>(t, param);
    else if (kind == 2) m = std::make_shared<Dielectric>(param);
    else if (kind == 3) m = std::make_shared<DiffuseLight>(t);
    else return -1;
    for (int k = 0; k < n; ++k) {
        HitRec h;
        h.t = 1.0f;
        h.u = h.v = 0.0f;
        h.p = Vector3(0.5f, 0.25f, -1.0f);
        h.n = Vector3(normals[3 * k], normals[3 * k + 1], normals[3 * k + 2]);
        h.mat = m;
        const Ray r(Vector3(0), Vector3(d_in[3 * k], d_in[3 * k + 1], d_in[3 * k + 2]));
        ScatterRec srec;
        srec.ray = Ray(Vector3(0), Vector3(0));
        srec.albedo = Vector3(0);
        srec.pdf = nullptr;
        srec.is_specular = false;
        rtx_rng_init(&g_rt_rng, uint32_t(k), 0u, seed);
        const bool did = m->scatter(r, h, srec);
        ok[k] = did ? 1 : 0;
        specular[k] = srec.is_specular ? 1 : 0;
        const Vector3 dd = srec.is_specular ? srec.ray.direction() : Vector3(0);
        const Vector3 e = m->emitted(r, h);
        const Ray sr(h.p, Vector3(scattered[3 * k], scattered[3 * k + 1], scattered[3 * k + 2]));
        spdf[k] = m->scattering_pdf(sr, h);
        for (int c = 0; c < 3; ++c) {
            dir[3 * k + c] = dd[c];
            albedo[3 * k + c] = kind == 3 ? tex[c] : srec.albedo[c];
            emitted[3 * k + c] = e[c];
        }
    }
    return 0;
}
}  // extern "C"
