// raytrace.h — host-side scene-building API of the MI355X path tracer.
//
// It keeps the class surface of the reference tracer (Raytrace_C++/Src): Vector3,
// Camera, the Shape hierarchy (Sphere, Rect, Box, ShapeList, FlipNormals,
// Translate, Rotate) with its fluent ShapeBuilder, the Material family
// (Lambertian, Metal, Dielectric, DiffuseLight), the Texture family
// (ColorTexture, CheckerTexture), Image and Scene — same names, constructor
// signatures and ownership conventions (shared_ptr for shapes / materials /
// textures), so scene code written for the reference compiles against it.
//
// Unlike the reference, these classes do not trace rays: they record their
// parameters and flatten themselves into the plain arrays of RtxScene
// (include/rtx.h); all intersection, scattering and integration happens on the
// GPU behind the C ABI.  There is no CPU rendering path.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "rtx.h"
#include "rtx_det.h"

namespace rt {

#ifndef PI
#define PI 3.14159265359f   // Util.h:2
#endif
#ifndef GAMMA_FACTOR
#define GAMMA_FACTOR 2.2f   // Util.h:8
#endif

// ---------------------------------------------------------------------------
// Vector3 / Quat: the subset of vectormath_aos.h the scene API needs, with the
// library's operation order (vec_aos.h:263-493, quat_aos.h:348-355).
// ---------------------------------------------------------------------------
class Vector3 {
public:
    Vector3() : mX(0), mY(0), mZ(0) {}
    Vector3(float x, float y, float z) : mX(x), mY(y), mZ(z) {}
    explicit Vector3(float s) : mX(s), mY(s), mZ(s) {}
    float getX() const { return mX; }
    float getY() const { return mY; }
    float getZ() const { return mZ; }
    float operator[](int i) const { return i == 0 ? mX : (i == 1 ? mY : mZ); }
    static Vector3 xAxis() { return Vector3(1.0f, 0.0f, 0.0f); }
    static Vector3 yAxis() { return Vector3(0.0f, 1.0f, 0.0f); }
    static Vector3 zAxis() { return Vector3(0.0f, 0.0f, 1.0f); }
    Vector3 operator+(const Vector3& v) const { return Vector3(mX + v.mX, mY + v.mY, mZ + v.mZ); }
    Vector3 operator-(const Vector3& v) const { return Vector3(mX - v.mX, mY - v.mY, mZ - v.mZ); }
    Vector3 operator-() const { return Vector3(-mX, -mY, -mZ); }
    Vector3 operator*(float s) const { return Vector3(mX * s, mY * s, mZ * s); }
    Vector3 operator/(float s) const { return Vector3(mX / s, mY / s, mZ / s); }

private:
    float mX, mY, mZ;
};
inline Vector3 operator*(float s, const Vector3& v) { return v * s; }
// vec_aos.h:92-95
inline Vector3 lerp(float t, const Vector3& a, const Vector3& b) { return a + ((b - a) * t); }
inline float dot(const Vector3& a, const Vector3& b) {
    float r = a.getX() * b.getX();
    r = r + a.getY() * b.getY();
    r = r + a.getZ() * b.getZ();
    return r;
}
inline float lengthSqr(const Vector3& v) { return dot(v, v); }
float length(const Vector3& v);
Vector3 normalize(const Vector3& v);
inline Vector3 cross(const Vector3& a, const Vector3& b) {
    return Vector3(a.getY() * b.getZ() - a.getZ() * b.getY(), a.getZ() * b.getX() - a.getX() * b.getZ(),
                   a.getX() * b.getY() - a.getY() * b.getX());
}
inline float radians(float deg) { return (deg / 180.f) * PI; }  // Util.h:24

class Quat {
public:
    Quat() : x(0), y(0), z(0), w(1) {}
    Quat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    // quat_aos.h:348-355; sin / cos are the bit-reproducible ones of rtx_det.h
    static Quat rotation(float radians, const Vector3& unitVec);
    float x, y, z, w;
};

// ---------------------------------------------------------------------------
// flattening
// ---------------------------------------------------------------------------
class Material;
class Shape;
class Texture;
typedef std::shared_ptr<Material> MaterialPtr;  // Pch.h:11-18
typedef std::shared_ptr<Shape> ShapePtr;
typedef std::shared_ptr<Texture> TexturePtr;

class Camera;

// Owns the arrays an RtxScene points into.
class FlatScene {
public:
    FlatScene();
    // The view is valid while this object is alive and unmodified.
    const RtxScene* view();

    struct StepRec { uint32_t kind; float x, y, z, w; };
    // wrapper state while walking the shape tree
    struct Cursor {
        std::vector<StepRec> steps;  // outermost first
        int chain = 0;               // interned chain id, -1 = not yet interned
        bool flip = false;
    };

    uint32_t add_prim(uint32_t kind, float a, float b, float c, float d, float e, const MaterialPtr& m,
                      Cursor& cur);
    // Light entries.  The entries of one list are contiguous: reserve them first, then fill
    // each slot (a nested list reserves its own children behind them).
    uint32_t reserve_lights(uint32_t n);
    void set_light(uint32_t slot, uint32_t kind, float a, float b, float c, float d, float e, Cursor& cur);
    void set_light_list(uint32_t slot, uint32_t first, uint32_t count);
    // RTX_LIGHTS_GENERAL: rects of every axis and wrapped shapes are sampled as lights
    void set_general_lights(bool g) { m_general_lights = g; }
    bool general_lights() const { return m_general_lights; }
    uint32_t intern_material(const MaterialPtr& m);
    uint32_t intern_texture(const TexturePtr& t);
    uint32_t new_texture(uint32_t kind, float r, float g, float b, uint32_t odd, uint32_t even);
    // 8-bit RGB image texture; returns the texture index
    uint32_t new_image_texture(int width, int height, const unsigned char* rgb);
    uint32_t new_material(uint32_t kind, uint32_t tex, float param);
    void set_camera(const Camera& cam);
    void set_background(const Vector3& c);
    void set_background_sky(bool sky) { m_background_sky = sky; }
    void set_light_is_list(bool v) { m_light_is_list = v; }
    const std::string& error() const { return m_error; }
    void fail(const std::string& e) { if (m_error.empty()) m_error = e; }

    std::vector<uint32_t> prim_kind, prim_flip, prim_material, prim_chain;
    std::vector<float> prim_a, prim_b, prim_c, prim_d, prim_e;
    std::vector<uint32_t> chain_begin, step_kind;
    std::vector<float> step_x, step_y, step_z, step_w;
    std::vector<uint32_t> mat_kind, mat_texture;
    std::vector<float> mat_param;
    std::vector<uint32_t> tex_kind, tex_odd, tex_even, tex_offset;
    std::vector<uint8_t> texels;
    std::vector<float> tex_r, tex_g, tex_b;
    std::vector<uint32_t> light_kind, light_first, light_count, light_chain;
    std::vector<float> light_a, light_b, light_c, light_d, light_e;
    uint32_t n_top_lights = 0;

private:
    int intern_chain(Cursor& cur);
    RtxScene m_view;
    bool m_light_is_list = true;
    bool m_general_lights = false;
    float m_cam[4][3];
    float m_background[3];
    bool m_background_sky = false;
    std::map<const Material*, uint32_t> m_mat_ids;
    std::map<const Texture*, uint32_t> m_tex_ids;
    std::string m_error;
};

// ---------------------------------------------------------------------------
// textures (Texture.h, ColorTexture.h, CheckerTexture.h)
// ---------------------------------------------------------------------------
class Texture {
public:
    virtual ~Texture() {}
    virtual uint32_t flatten(FlatScene& f) const = 0;
};

class ColorTexture : public Texture {
public:
    ColorTexture(const Vector3& c) : m_color(c) {}
    uint32_t flatten(FlatScene& f) const override;

private:
    Vector3 m_color;
};

// No counterpart among the reference's texture classes: value(u, v, p) = Scene::backgroundSky(p - centre)
// (Scene.h:24-28).  On an emitting sphere around the scene it turns the sky gradient into a
// surface that Scene::color can render and sample as a light.
class SkyTexture : public Texture {
public:
    SkyTexture(const Vector3& centre) : m_centre(centre) {}
    uint32_t flatten(FlatScene& f) const override;

private:
    Vector3 m_centre;
};

class CheckerTexture : public Texture {
public:
    CheckerTexture(const TexturePtr& t0, const TexturePtr& t1, float freq) : m_odd(t0), m_even(t1), m_freq(freq) {}
    uint32_t flatten(FlatScene& f) const override;

private:
    TexturePtr m_odd, m_even;
    float m_freq;
};

// ImageTexture.h: nearest-texel lookup into an 8-bit RGB image.  The reference decodes the
// file with stb_image (JPEG, PNG, ...); this class reads 24-bit uncompressed BMP and binary
// PPM (P6) files, or takes the texels directly.
class ImageTexture : public Texture {
public:
    ImageTexture(const char* name);
    ImageTexture(int width, int height, const unsigned char* rgb);
    uint32_t flatten(FlatScene& f) const override;
    int width() const { return m_width; }
    int height() const { return m_height; }
    bool loaded() const { return !m_texels.empty(); }
    const unsigned char* texels() const { return m_texels.data(); }  // rows top-down, RGB
    const std::string& error() const { return m_error; }             // why a JPEG file was refused

private:
    int m_width = 0, m_height = 0;
    std::vector<unsigned char> m_texels;  // rows top-down, RGB
    std::string m_name, m_error;
};

// ---------------------------------------------------------------------------
// materials (Material.h, Lambertian.h, Metal.h, Dielectric.h, DiffuseLight.h)
// ---------------------------------------------------------------------------
class Material {
public:
    virtual ~Material() {}
    virtual uint32_t flatten(FlatScene& f) const = 0;
    virtual void set_texture(const TexturePtr&) {}
};

class Lambertian : public Material {
public:
    Lambertian(const TexturePtr& a) : m_albedo(a) {}
    uint32_t flatten(FlatScene& f) const override;
    void set_texture(const TexturePtr& a) override { m_albedo = a; }

private:
    TexturePtr m_albedo;
};

class Metal : public Material {
public:
    Metal(const TexturePtr& a, float fuzz) : m_albedo(a), m_fuzz(fuzz) {}
    uint32_t flatten(FlatScene& f) const override;
    void set_texture(const TexturePtr& a) override { m_albedo = a; }
    void set_fuzz(float fuzz) { m_fuzz = fuzz; }

private:
    TexturePtr m_albedo;
    float m_fuzz;
};

class Dielectric : public Material {
public:
    Dielectric(float ri) : m_ri(ri) {}
    uint32_t flatten(FlatScene& f) const override;

private:
    float m_ri;
};

class DiffuseLight : public Material {
public:
    DiffuseLight(const TexturePtr& emit) : m_emit(emit) {}
    uint32_t flatten(FlatScene& f) const override;

private:
    TexturePtr m_emit;
};

// ---------------------------------------------------------------------------
// shapes (Shape.h, Sphere.h, Rect.h, Box.h, ShapeList.h, FlipNormals.h,
// Translate.h, Rotate.h)
// ---------------------------------------------------------------------------
class Shape {
public:
    virtual ~Shape() {}
    // Emit this shape's leaves, in the order the reference's hit() visits them.
    virtual void flatten(FlatScene& f, FlatScene::Cursor& cur) const = 0;
    // Emit this shape as entry `slot` of the light arrays.  Only Rect, Sphere and ShapeList
    // override pdf_value() / random() in the reference (Shape.h:8-9); everything else keeps
    // the defaults, unless the scene asks for general light sampling, in which case the
    // wrappers pass the query on to the shape they wrap (`cur` carries their transforms).
    virtual void flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const {
        f.set_light(slot, RTX_LIGHT_OTHER, 0, 0, 0, 0, 0, cur);
    }
    virtual bool is_list() const { return false; }
};

class Sphere : public Shape {
public:
    Sphere() : m_radius(0) {}
    Sphere(const Vector3& c, float r, const MaterialPtr& mat) : m_center(c), m_radius(r), m_material(mat) {}
    void flatten(FlatScene& f, FlatScene::Cursor& cur) const override;
    void flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const override;

private:
    Vector3 m_center;
    float m_radius;
    MaterialPtr m_material;
};

class Rect : public Shape {
public:
    enum AxisType { kXY = 0, kXZ, kYZ };
    Rect() : m_x0(0), m_x1(0), m_y0(0), m_y1(0), m_k(0), m_axis(kXY) {}
    Rect(float x0, float x1, float y0, float y1, float k, AxisType axis, const MaterialPtr& m)
        : m_x0(x0), m_x1(x1), m_y0(y0), m_y1(y1), m_k(k), m_axis(axis), m_material(m) {}
    void flatten(FlatScene& f, FlatScene::Cursor& cur) const override;
    void flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const override;

private:
    float m_x0, m_x1, m_y0, m_y1, m_k;
    AxisType m_axis;
    MaterialPtr m_material;
};

class ShapeList : public Shape {
public:
    ShapeList() {}
    void add(const ShapePtr& shape) { m_list.push_back(shape); }
    size_t size() const { return m_list.size(); }
    void flatten(FlatScene& f, FlatScene::Cursor& cur) const override;
    // a list nested inside the light shape: one RTX_LIGHT_LIST entry + its children
    void flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const override;
    bool is_list() const override { return true; }
    // Emit the entries of this list as the scene's top-level light list.
    void flatten_as_light_list(FlatScene& f) const;

private:
    std::vector<ShapePtr> m_list;
};

class Box : public Shape {
public:
    Box() {}
    Box(const Vector3& p0, const Vector3& p1, const MaterialPtr& m);
    void flatten(FlatScene& f, FlatScene::Cursor& cur) const override;
    void flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const override;

private:
    Vector3 m_p0, m_p1;
    std::unique_ptr<ShapeList> m_list;
};

class FlipNormals : public Shape {
public:
    FlipNormals(const ShapePtr& shape) : m_shape(shape) {}
    void flatten(FlatScene& f, FlatScene::Cursor& cur) const override;
    void flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const override;

private:
    ShapePtr m_shape;
};

class Translate : public Shape {
public:
    Translate(const ShapePtr& sp, const Vector3& displacement) : m_shape(sp), m_offset(displacement) {}
    void flatten(FlatScene& f, FlatScene::Cursor& cur) const override;
    void flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const override;

private:
    ShapePtr m_shape;
    Vector3 m_offset;
};

class Rotate : public Shape {
public:
    Rotate(const ShapePtr& sp, const Vector3& axis, float angle)
        : m_shape(sp), m_quat(Quat::rotation(radians(angle), axis)) {}
    void flatten(FlatScene& f, FlatScene::Cursor& cur) const override;
    void flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const override;

private:
    ShapePtr m_shape;
    Quat m_quat;
};

// Fluent factory, ShapeBuilder.h:10-95.
class ShapeBuilder {
public:
    ShapeBuilder() {}
    ShapeBuilder(const ShapePtr& sp) : m_ptr(sp) {}
    ShapeBuilder& reset(const ShapePtr& sp) { m_ptr = sp; return *this; }
    ShapeBuilder& sphere(const Vector3& c, float r, const MaterialPtr& m);
    ShapeBuilder& rect(float x0, float x1, float y0, float y1, float k, Rect::AxisType axis, const MaterialPtr& m);
    ShapeBuilder& rectXY(float x0, float x1, float y0, float y1, float k, const MaterialPtr& m);
    ShapeBuilder& rectXZ(float x0, float x1, float y0, float y1, float k, const MaterialPtr& m);
    ShapeBuilder& rectYZ(float x0, float x1, float y0, float y1, float k, const MaterialPtr& m);
    // corner-point overloads: the two in-plane coordinates of p0 / p1 are used
    ShapeBuilder& rect(const Vector3& p0, const Vector3& p1, float k, Rect::AxisType axis, const MaterialPtr& m);
    ShapeBuilder& rectXY(const Vector3& p0, const Vector3& p1, float k, const MaterialPtr& m);
    ShapeBuilder& rectXZ(const Vector3& p0, const Vector3& p1, float k, const MaterialPtr& m);
    ShapeBuilder& rectYZ(const Vector3& p0, const Vector3& p1, float k, const MaterialPtr& m);
    ShapeBuilder& box(const Vector3& p0, const Vector3& p1, const MaterialPtr& m);
    ShapeBuilder& flip();
    ShapeBuilder& translate(const Vector3& t);
    ShapeBuilder& rotate(const Vector3& axis, float angle);
    const ShapePtr& get() const { return m_ptr; }

private:
    ShapePtr m_ptr;
};

// ---------------------------------------------------------------------------
// camera (Camera.h)
// ---------------------------------------------------------------------------
class Camera {
public:
    Camera() {}
    Camera(const Vector3& u, const Vector3& v, const Vector3& w);
    Camera(const Vector3& lookfrom, const Vector3& lookat, const Vector3& vup, float vfov, float aspect);
    const Vector3& origin() const { return m_origin; }
    const Vector3& horizontal() const { return m_uvw[0]; }
    const Vector3& vertical() const { return m_uvw[1]; }
    const Vector3& lowerLeft() const { return m_uvw[2]; }

private:
    Vector3 m_origin;
    Vector3 m_uvw[3];
};

// ---------------------------------------------------------------------------
// image (Image.h, DenanFilter.h, GammaFilter.h, TonemapFilter.h)
// ---------------------------------------------------------------------------
class Image {
public:
    struct rgb { unsigned char r, g, b; };
    Image() : m_width(0), m_height(0) {}
    Image(int w, int h);
    int width() const { return m_width; }
    int height() const { return m_height; }
    void* pixels() const { return const_cast<rgb*>(m_pixels.data()); }
    // NaN filter, gamma 1/2.2, clamp, *255.99 truncated (Image.h:30-39)
    void write(int x, int y, float r, float g, float b);
    // 24-bit bottom-up BMP, the format stbi_write_bmp produces (Scene.cpp:155)
    bool save_bmp(const char* path) const;

private:
    int m_width, m_height;
    std::vector<rgb> m_pixels;
};

// ---------------------------------------------------------------------------
// scene (Scene.h)
// ---------------------------------------------------------------------------
class Scene {
public:
    Scene(const char* fileName, int width, int height, int sample);
    virtual ~Scene();

    // Defines camera, world and light list.  The default is the Cornell-box variant
    // of the reference (Scene.cpp:37-87); override it or use the setters below.
    virtual void build();
    // build(), render on the GPU, write the BMP.  Returns false (see lastError())
    // on failure; the reference's render() returns void and reports nothing.
    bool render();
    const char* getFilename() const { return m_filename.c_str(); }

    void setCamera(Camera* cam) { m_camera.reset(cam); }
    void setWorld(Shape* world) { m_world.reset(world); }
    void setLight(Shape* light) { m_light.reset(light); }
    void setBackground(const Vector3& c) { m_backColor = c; }
    Vector3 background(const Vector3&) const { return m_backColor; }
    // Scene.h:24-28: the vertical white-to-blue gradient; the reference defines it but its
    // color() never calls it.  setBackgroundSky(true) makes misses return it.
    Vector3 backgroundSky(const Vector3& d) const;
    void setBackgroundSky(bool sky) { m_backgroundSky = sky; }
    // An extension: sample rects of every axis and flipped / translated / rotated shapes and
    // boxes as lights.  The reference gives all of these the defaults of Shape.h:8-9 (density 0,
    // direction (1,0,0)), which stays the behaviour unless this is switched on.
    void setGeneralLightSampling(bool g) { m_generalLights = g; }

    // render options the reference fixes at compile time
    void setMaxDepth(int d) { m_maxDepth = d; }          // MAX_DEPTH, Scene.cpp:9
    void setSeed(uint64_t s) { m_seed = s; }
    void setDeterministic(bool d) { m_deterministic = d; }
    void setDevice(int d) { m_device = d; }
    // Several GPUs of this host: each renders a contiguous block of every launch's sample range
    // and the partial frames are summed on the first one (rtx_render_multi).  An empty list
    // (the default) means the single device of setDevice().
    void setDevices(const std::vector<int>& d) { m_devices = d; }
    void setSamplesPerLaunch(int s) { m_samplesPerLaunch = s; }
    // Whether the GPU library compiles its fast-mode kernels for this scene (RTX_OPTION_SPECIALIZE_SCAN
    // of include/rtx.h; scenes of at most 64 primitives): Never, AtUpload (render() then takes 0.3 - 0.7 s
    // longer to start and its launches run 7 - 22 % faster), InBackground (render() starts at once and
    // switches kernels when they are ready: two runs may then differ in the rounding of some samples), or
    // Auto, the default: AtUpload for a frame of at least kCompileFromSamples pixel samples - where the
    // compilation is paid back by the launches - and Never below.
    enum class KernelCompilation { Auto = -1, Never = 0, AtUpload = 1, InBackground = 2 };
    static constexpr unsigned long long kCompileFromSamples = 120000000000ull;  // 1080p x 58 000 spp: about 3 s of launches
    void setKernelCompilation(KernelCompilation m) { m_kernelCompilation = m; }

    // Flatten camera / world / light into `out`.  False on a scene the GPU path
    // cannot express (see FlatScene::error()).
    bool flatten(FlatScene& out) const;
    const Image& image() const { return *m_image; }
    const std::string& lastError() const { return m_error; }
    unsigned long long raysTraced() const { return m_rays; }
    double renderSeconds() const { return m_seconds; }

protected:
    std::unique_ptr<Camera> m_camera;
    std::unique_ptr<Image> m_image;
    Vector3 m_backColor;
    std::string m_filename;
    std::unique_ptr<Shape> m_world;
    int m_samples;
    std::unique_ptr<Shape> m_light;
    bool m_backgroundSky = false;
    bool m_generalLights = false;
    int m_maxDepth = 50;
    uint64_t m_seed = 0;
    bool m_deterministic = false;
    int m_device = 0;
    std::vector<int> m_devices;
    // per device.  Each rtx_render call also moves the frame over PCIe (once, twice when it adds to earlier calls) and
    // ends on its slowest waves: 1080p x 4 096 spp through calls of 512 / 1 024 / 4 096 samples runs at 112 / 122 /
    // 125 Grays/s end to end
    int m_samplesPerLaunch = 4096;
    KernelCompilation m_kernelCompilation = KernelCompilation::Auto;
    std::string m_error;
    unsigned long long m_rays = 0;
    double m_seconds = 0.0;
};

}  // namespace rt
