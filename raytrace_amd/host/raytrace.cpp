// raytrace.cpp — implementation of the host scene API: flattening of the shape /
// material / texture graph into RtxScene arrays, the image filter chain, the BMP
// writer and Scene::render(), which drives the GPU through the C ABI.
#include "raytrace.h"

#include "image_decode.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace rt {

// ---------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------
float length(const Vector3& v) { return ::sqrtf(lengthSqr(v)); }

Vector3 normalize(const Vector3& v) {  // vec_aos.h:474-485
    float lenSqr = lengthSqr(v);
    float lenInv = 1.0f / ::sqrtf(lenSqr);
    return Vector3(v.getX() * lenInv, v.getY() * lenInv, v.getZ() * lenInv);
}

Quat Quat::rotation(float radians, const Vector3& unitVec) {  // quat_aos.h:348-355
    float angle = radians * 0.5f;
    float s = rtx_sinf(angle);
    float c = rtx_cosf(angle);
    Vector3 v = unitVec * s;
    return Quat(v.getX(), v.getY(), v.getZ(), c);
}

// ---------------------------------------------------------------------------
// camera, Camera.h:5-22
// ---------------------------------------------------------------------------
Camera::Camera(const Vector3& u, const Vector3& v, const Vector3& w) {
    m_origin = Vector3(0.0f);
    m_uvw[0] = u;
    m_uvw[1] = v;
    m_uvw[2] = w;
}

Camera::Camera(const Vector3& lookfrom, const Vector3& lookat, const Vector3& vup, float vfov, float aspect) {
    float halfH = rtx_tanf(radians(vfov) / 2.0f);
    float halfW = aspect * halfH;
    m_origin = lookfrom;
    Vector3 w = normalize(lookfrom - lookat);
    Vector3 u = normalize(cross(vup, w));
    Vector3 v = cross(w, u);
    m_uvw[2] = m_origin - halfW * u - halfH * v - w;
    m_uvw[0] = 2.0f * halfW * u;
    m_uvw[1] = 2.0f * halfH * v;
}

// ---------------------------------------------------------------------------
// FlatScene
// ---------------------------------------------------------------------------
FlatScene::FlatScene() {
    std::memset(&m_view, 0, sizeof m_view);
    std::memset(m_cam, 0, sizeof m_cam);
    std::memset(m_background, 0, sizeof m_background);
    chain_begin.push_back(0);  // chain 0: no steps
    chain_begin.push_back(0);
}

int FlatScene::intern_chain(Cursor& cur) {
    if (cur.chain >= 0) return cur.chain;
    if (cur.steps.empty()) return cur.chain = 0;
    for (const StepRec& s : cur.steps) {
        step_kind.push_back(s.kind);
        step_x.push_back(s.x);
        step_y.push_back(s.y);
        step_z.push_back(s.z);
        step_w.push_back(s.w);
    }
    chain_begin.push_back((uint32_t)step_kind.size());
    return cur.chain = (int)chain_begin.size() - 2;
}

uint32_t FlatScene::add_prim(uint32_t kind, float a, float b, float c, float d, float e, const MaterialPtr& m,
                             Cursor& cur) {
    if (!m) {
        fail("a world primitive has no material (the reference would dereference a null MaterialPtr)");
        return 0;
    }
    prim_kind.push_back(kind);
    prim_flip.push_back(cur.flip ? 1u : 0u);
    prim_material.push_back(intern_material(m));
    prim_chain.push_back((uint32_t)intern_chain(cur));
    prim_a.push_back(a);
    prim_b.push_back(b);
    prim_c.push_back(c);
    prim_d.push_back(d);
    prim_e.push_back(e);
    return (uint32_t)prim_kind.size() - 1;
}

uint32_t FlatScene::reserve_lights(uint32_t n) {
    const uint32_t first = (uint32_t)light_kind.size();
    light_kind.resize(first + n, RTX_LIGHT_OTHER);
    for (std::vector<uint32_t>* v : {&light_first, &light_count, &light_chain}) v->resize(first + n, 0u);
    for (std::vector<float>* v : {&light_a, &light_b, &light_c, &light_d, &light_e}) v->resize(first + n, 0.0f);
    return first;
}

void FlatScene::set_light(uint32_t slot, uint32_t kind, float a, float b, float c, float d, float e, Cursor& cur) {
    light_kind[slot] = kind;
    light_a[slot] = a;
    light_b[slot] = b;
    light_c[slot] = c;
    light_d[slot] = d;
    light_e[slot] = e;
    // only general light sampling looks at a light's transforms (and only it lets them be recorded)
    light_chain[slot] = m_general_lights && kind != RTX_LIGHT_OTHER ? (uint32_t)intern_chain(cur) : 0u;
}

void FlatScene::set_light_list(uint32_t slot, uint32_t first, uint32_t count) {
    light_kind[slot] = RTX_LIGHT_LIST;
    light_first[slot] = first;
    light_count[slot] = count;
}

uint32_t FlatScene::intern_material(const MaterialPtr& m) {
    auto it = m_mat_ids.find(m.get());
    if (it != m_mat_ids.end()) return it->second;
    uint32_t id = m->flatten(*this);
    m_mat_ids[m.get()] = id;
    return id;
}

uint32_t FlatScene::intern_texture(const TexturePtr& t) {
    if (!t) {
        fail("a material has no texture");
        return 0;
    }
    auto it = m_tex_ids.find(t.get());
    if (it != m_tex_ids.end()) return it->second;
    uint32_t id = t->flatten(*this);
    m_tex_ids[t.get()] = id;
    return id;
}

uint32_t FlatScene::new_texture(uint32_t kind, float r, float g, float b, uint32_t odd, uint32_t even) {
    tex_kind.push_back(kind);
    tex_r.push_back(r);
    tex_g.push_back(g);
    tex_b.push_back(b);
    tex_odd.push_back(odd);
    tex_even.push_back(even);
    tex_offset.push_back(0);
    return (uint32_t)tex_kind.size() - 1;
}

uint32_t FlatScene::new_image_texture(int width, int height, const unsigned char* rgb) {
    const uint32_t id = new_texture(RTX_TEX_IMAGE, 0.0f, 0.0f, 0.0f, (uint32_t)width, (uint32_t)height);
    tex_offset[id] = (uint32_t)texels.size();
    texels.insert(texels.end(), rgb, rgb + (size_t)width * height * 3);
    return id;
}

uint32_t FlatScene::new_material(uint32_t kind, uint32_t tex, float param) {
    mat_kind.push_back(kind);
    mat_texture.push_back(tex);
    mat_param.push_back(param);
    return (uint32_t)mat_kind.size() - 1;
}

void FlatScene::set_camera(const Camera& cam) {
    const Vector3* v[4] = {&cam.origin(), &cam.lowerLeft(), &cam.horizontal(), &cam.vertical()};
    for (int k = 0; k < 4; ++k) {
        m_cam[k][0] = v[k]->getX();
        m_cam[k][1] = v[k]->getY();
        m_cam[k][2] = v[k]->getZ();
    }
}

void FlatScene::set_background(const Vector3& c) {
    m_background[0] = c.getX();
    m_background[1] = c.getY();
    m_background[2] = c.getZ();
}

const RtxScene* FlatScene::view() {
    RtxScene& s = m_view;
    s.abi_version = RTX_ABI_VERSION;
    s.n_prims = (uint32_t)prim_kind.size();
    s.prim_kind = prim_kind.data();
    s.prim_flip = prim_flip.data();
    s.prim_material = prim_material.data();
    s.prim_chain = prim_chain.data();
    s.prim_a = prim_a.data();
    s.prim_b = prim_b.data();
    s.prim_c = prim_c.data();
    s.prim_d = prim_d.data();
    s.prim_e = prim_e.data();
    s.n_chains = (uint32_t)chain_begin.size() - 1;
    s.chain_begin = chain_begin.data();
    s.n_steps = (uint32_t)step_kind.size();
    s.step_kind = step_kind.data();
    s.step_x = step_x.data();
    s.step_y = step_y.data();
    s.step_z = step_z.data();
    s.step_w = step_w.data();
    s.n_materials = (uint32_t)mat_kind.size();
    s.mat_kind = mat_kind.data();
    s.mat_texture = mat_texture.data();
    s.mat_param = mat_param.data();
    s.n_textures = (uint32_t)tex_kind.size();
    s.tex_kind = tex_kind.data();
    s.tex_r = tex_r.data();
    s.tex_g = tex_g.data();
    s.tex_b = tex_b.data();
    s.tex_odd = tex_odd.data();
    s.tex_even = tex_even.data();
    s.tex_offset = tex_offset.data();
    s.n_texel_bytes = texels.size();
    s.texels = texels.data();
    s.n_lights = n_top_lights;
    s.n_light_entries = (uint32_t)light_kind.size();
    s.light_sampling = m_general_lights ? RTX_LIGHTS_GENERAL : RTX_LIGHTS_REFERENCE;
    s.light_first = light_first.data();
    s.light_count = light_count.data();
    s.light_chain = light_chain.data();
    s.light_is_list = m_light_is_list ? 1u : 0u;
    s.light_kind = light_kind.data();
    s.light_a = light_a.data();
    s.light_b = light_b.data();
    s.light_c = light_c.data();
    s.light_d = light_d.data();
    s.light_e = light_e.data();
    std::memcpy(s.cam_origin, m_cam[0], sizeof s.cam_origin);
    std::memcpy(s.cam_ll, m_cam[1], sizeof s.cam_ll);
    std::memcpy(s.cam_u, m_cam[2], sizeof s.cam_u);
    std::memcpy(s.cam_v, m_cam[3], sizeof s.cam_v);
    std::memcpy(s.background, m_background, sizeof s.background);
    s.background_mode = m_background_sky ? RTX_BACKGROUND_SKY : RTX_BACKGROUND_CONSTANT;
    return &s;
}

// ---------------------------------------------------------------------------
// textures and materials
// ---------------------------------------------------------------------------
uint32_t ColorTexture::flatten(FlatScene& f) const {
    return f.new_texture(RTX_TEX_COLOR, m_color.getX(), m_color.getY(), m_color.getZ(), 0, 0);
}

uint32_t SkyTexture::flatten(FlatScene& f) const {
    return f.new_texture(RTX_TEX_SKY, m_centre.getX(), m_centre.getY(), m_centre.getZ(), 0, 0);
}

uint32_t CheckerTexture::flatten(FlatScene& f) const {
    uint32_t odd = f.intern_texture(m_odd);
    uint32_t even = f.intern_texture(m_even);
    return f.new_texture(RTX_TEX_CHECKER, m_freq, 0.0f, 0.0f, odd, even);
}

// ---------------------------------------------------------------------------
// image textures
// ---------------------------------------------------------------------------
namespace {
uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

bool load_bmp(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgb) {
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') return false;
    const uint32_t off = rd32(&d[10]);
    const int32_t bw = (int32_t)rd32(&d[18]), bh = (int32_t)rd32(&d[22]);
    const uint32_t bpp = d[28] | (d[29] << 8), compression = rd32(&d[30]);
    if (bpp != 24 || compression != 0 || bw <= 0 || bh == 0) return false;
    w = bw;
    h = bh < 0 ? -bh : bh;
    const size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
    if (d.size() < off + stride * h) return false;
    rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) {
        const int src_row = bh > 0 ? h - 1 - y : y;  // positive height: bottom-up file
        const unsigned char* src = &d[off + stride * src_row];
        for (int x = 0; x < w; ++x) {
            rgb[((size_t)y * w + x) * 3 + 0] = src[3 * x + 2];
            rgb[((size_t)y * w + x) * 3 + 1] = src[3 * x + 1];
            rgb[((size_t)y * w + x) * 3 + 2] = src[3 * x + 0];
        }
    }
    return true;
}

bool load_ppm(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgb) {
    if (d.size() < 11 || d[0] != 'P' || d[1] != '6') return false;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < d.size()) {
        while (pos < d.size() && (d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\r' || d[pos] == '\t')) ++pos;
        if (pos < d.size() && d[pos] == '#') { while (pos < d.size() && d[pos] != '\n') ++pos; continue; }
        int v = 0;
        bool any = false;
        while (pos < d.size() && d[pos] >= '0' && d[pos] <= '9') { v = v * 10 + (d[pos] - '0'); ++pos; any = true; }
        if (!any) return false;
        vals[got++] = v;
    }
    if (got < 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0) return false;
    ++pos;  // the single whitespace after maxval
    w = vals[0];
    h = vals[1];
    if (d.size() < pos + (size_t)w * h * 3) return false;
    rgb.assign(d.begin() + pos, d.begin() + pos + (size_t)w * h * 3);
    return true;
}
}  // namespace

ImageTexture::ImageTexture(const char* name) : m_name(name ? name : "") {
    FILE* fp = name ? std::fopen(name, "rb") : nullptr;
    if (!fp) return;
    std::vector<unsigned char> data;
    unsigned char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, fp)) > 0) data.insert(data.end(), buf, buf + n);
    std::fclose(fp);
    if (load_bmp(data, m_width, m_height, m_texels) || load_ppm(data, m_width, m_height, m_texels)) return;
    if (data.size() >= 2 && data[0] == 0xFF && data[1] == 0xD8 &&
        detail::load_jpeg(data, m_width, m_height, m_texels, m_error))
        return;
    m_width = m_height = 0;
    m_texels.clear();
}

ImageTexture::ImageTexture(int width, int height, const unsigned char* rgb)
    : m_width(width), m_height(height), m_texels(rgb, rgb + (size_t)width * height * 3), m_name("<memory>") {}

uint32_t ImageTexture::flatten(FlatScene& f) const {
    if (!loaded()) {
        f.fail("ImageTexture '" + m_name + "': cannot read (24-bit BMP, binary PPM and sequential JPEG are supported)" +
               (m_error.empty() ? std::string() : ": " + m_error));
        return 0;
    }
    return f.new_image_texture(m_width, m_height, m_texels.data());
}

uint32_t Lambertian::flatten(FlatScene& f) const {
    return f.new_material(RTX_MAT_LAMBERTIAN, f.intern_texture(m_albedo), 0.0f);
}
uint32_t Metal::flatten(FlatScene& f) const {
    return f.new_material(RTX_MAT_METAL, f.intern_texture(m_albedo), m_fuzz);
}
uint32_t Dielectric::flatten(FlatScene& f) const { return f.new_material(RTX_MAT_DIELECTRIC, 0, m_ri); }
uint32_t DiffuseLight::flatten(FlatScene& f) const {
    return f.new_material(RTX_MAT_DIFFUSE_LIGHT, f.intern_texture(m_emit), 0.0f);
}

// ---------------------------------------------------------------------------
// shapes
// ---------------------------------------------------------------------------
void Sphere::flatten(FlatScene& f, FlatScene::Cursor& cur) const {
    f.add_prim(RTX_PRIM_SPHERE, m_center.getX(), m_center.getY(), m_center.getZ(), m_radius, 0.0f, m_material, cur);
}
void Sphere::flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const {
    f.set_light(slot, RTX_LIGHT_SPHERE, m_center.getX(), m_center.getY(), m_center.getZ(), m_radius, 0.0f, cur);
}

void Rect::flatten(FlatScene& f, FlatScene::Cursor& cur) const {
    f.add_prim((uint32_t)m_axis, m_x0, m_x1, m_y0, m_y1, m_k, m_material, cur);
}
void Rect::flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const {
    // Rect::pdf_value / random only act for kXZ (Rect.cpp:37,51); unless the scene asks for
    // general light sampling the kernel treats the other two axes like any shape that keeps
    // the defaults.
    f.set_light(slot, (uint32_t)m_axis, m_x0, m_x1, m_y0, m_y1, m_k, cur);
}

void ShapeList::flatten(FlatScene& f, FlatScene::Cursor& cur) const {
    for (const ShapePtr& p : m_list) {
        if (!p) { f.fail("ShapeList holds a null shape"); continue; }
        p->flatten(f, cur);
    }
}
void ShapeList::flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const {
    if (m_list.empty()) {
        f.fail("an empty ShapeList inside the light shape (undefined behaviour in the reference, ShapeList.cpp:28-35)");
        return;
    }
    const uint32_t first = f.reserve_lights((uint32_t)m_list.size());
    f.set_light_list(slot, first, (uint32_t)m_list.size());
    for (size_t k = 0; k < m_list.size(); ++k) {
        if (!m_list[k]) { f.fail("light list holds a null shape"); continue; }
        m_list[k]->flatten_light(f, first + (uint32_t)k, cur);
    }
}
void ShapeList::flatten_as_light_list(FlatScene& f) const {
    const uint32_t first = f.reserve_lights((uint32_t)m_list.size());  // 0: the top-level entries come first
    f.n_top_lights = (uint32_t)m_list.size();
    FlatScene::Cursor cur;
    for (size_t k = 0; k < m_list.size(); ++k) {
        if (!m_list[k]) { f.fail("light list holds a null shape"); continue; }
        m_list[k]->flatten_light(f, first + (uint32_t)k, cur);
    }
}

Box::Box(const Vector3& p0, const Vector3& p1, const MaterialPtr& m)
    : m_p0(p0), m_p1(p1), m_list(new ShapeList()) {
    // six faces, +z -z +y -y +x -x, the far face of each pair facing outwards as is
    // and the near face flipped (Box.cpp:15-28)
    struct Face { float a0, a1, b0, b1, k_far, k_near; Rect::AxisType axis; };
    const Face faces[3] = {
        {p0.getX(), p1.getX(), p0.getY(), p1.getY(), p1.getZ(), p0.getZ(), Rect::kXY},
        {p0.getX(), p1.getX(), p0.getZ(), p1.getZ(), p1.getY(), p0.getY(), Rect::kXZ},
        {p0.getY(), p1.getY(), p0.getZ(), p1.getZ(), p1.getX(), p0.getX(), Rect::kYZ},
    };
    for (const Face& fc : faces) {
        m_list->add(std::make_shared<Rect>(fc.a0, fc.a1, fc.b0, fc.b1, fc.k_far, fc.axis, m));
        m_list->add(std::make_shared<FlipNormals>(
            std::make_shared<Rect>(fc.a0, fc.a1, fc.b0, fc.b1, fc.k_near, fc.axis, m)));
    }
}
void Box::flatten(FlatScene& f, FlatScene::Cursor& cur) const { m_list->flatten(f, cur); }
// Box, FlipNormals, Translate and Rotate keep Shape's default pdf_value() / random() in the
// reference (Box.h, FlipNormals.h, Translate.h, Rotate.h override hit() only).  With general
// light sampling they pass the query on: a box is the list of its six faces, a flip changes
// nothing (the densities use |cos|), a translation / rotation joins the leaf's chain.
void Box::flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const {
    if (f.general_lights()) m_list->flatten_light(f, slot, cur);
    else Shape::flatten_light(f, slot, cur);
}

void FlipNormals::flatten(FlatScene& f, FlatScene::Cursor& cur) const {
    if (!m_shape) { f.fail("FlipNormals wraps a null shape"); return; }
    // negation commutes exactly with every later rotation / translation of the
    // normal, so nested flips reduce to a parity
    cur.flip = !cur.flip;
    m_shape->flatten(f, cur);
    cur.flip = !cur.flip;
}

void FlipNormals::flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const {
    if (f.general_lights() && m_shape) m_shape->flatten_light(f, slot, cur);
    else Shape::flatten_light(f, slot, cur);
}

void Translate::flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const {
    if (!f.general_lights() || !m_shape) { Shape::flatten_light(f, slot, cur); return; }
    FlatScene::Cursor inner;
    inner.steps = cur.steps;
    inner.steps.push_back({RTX_STEP_TRANSLATE, m_offset.getX(), m_offset.getY(), m_offset.getZ(), 0.0f});
    inner.chain = -1;
    m_shape->flatten_light(f, slot, inner);
}

void Rotate::flatten_light(FlatScene& f, uint32_t slot, FlatScene::Cursor& cur) const {
    if (!f.general_lights() || !m_shape) { Shape::flatten_light(f, slot, cur); return; }
    FlatScene::Cursor inner;
    inner.steps = cur.steps;
    inner.steps.push_back({RTX_STEP_ROTATE, m_quat.x, m_quat.y, m_quat.z, m_quat.w});
    inner.chain = -1;
    m_shape->flatten_light(f, slot, inner);
}

void Translate::flatten(FlatScene& f, FlatScene::Cursor& cur) const {
    if (!m_shape) { f.fail("Translate wraps a null shape"); return; }
    FlatScene::Cursor inner;
    inner.steps = cur.steps;
    inner.steps.push_back({RTX_STEP_TRANSLATE, m_offset.getX(), m_offset.getY(), m_offset.getZ(), 0.0f});
    inner.chain = -1;
    inner.flip = cur.flip;
    m_shape->flatten(f, inner);
}

void Rotate::flatten(FlatScene& f, FlatScene::Cursor& cur) const {
    if (!m_shape) { f.fail("Rotate wraps a null shape"); return; }
    FlatScene::Cursor inner;
    inner.steps = cur.steps;
    inner.steps.push_back({RTX_STEP_ROTATE, m_quat.x, m_quat.y, m_quat.z, m_quat.w});
    inner.chain = -1;
    inner.flip = cur.flip;
    m_shape->flatten(f, inner);
}

// ---------------------------------------------------------------------------
// ShapeBuilder
// ---------------------------------------------------------------------------
ShapeBuilder& ShapeBuilder::sphere(const Vector3& c, float r, const MaterialPtr& m) {
    m_ptr = std::make_shared<Sphere>(c, r, m);
    return *this;
}
ShapeBuilder& ShapeBuilder::rect(float x0, float x1, float y0, float y1, float k, Rect::AxisType axis,
                                 const MaterialPtr& m) {
    m_ptr = std::make_shared<Rect>(x0, x1, y0, y1, k, axis, m);
    return *this;
}
ShapeBuilder& ShapeBuilder::rectXY(float x0, float x1, float y0, float y1, float k, const MaterialPtr& m) {
    return rect(x0, x1, y0, y1, k, Rect::kXY, m);
}
ShapeBuilder& ShapeBuilder::rectXZ(float x0, float x1, float y0, float y1, float k, const MaterialPtr& m) {
    return rect(x0, x1, y0, y1, k, Rect::kXZ, m);
}
ShapeBuilder& ShapeBuilder::rectYZ(float x0, float x1, float y0, float y1, float k, const MaterialPtr& m) {
    return rect(x0, x1, y0, y1, k, Rect::kYZ, m);
}
ShapeBuilder& ShapeBuilder::rect(const Vector3& p0, const Vector3& p1, float k, Rect::AxisType axis,
                                 const MaterialPtr& m) {
    // in-plane coordinate pair per axis type (ShapeBuilder.h:44-60)
    const int first = (axis == Rect::kYZ) ? 1 : 0;
    const int second = (axis == Rect::kXY) ? 1 : 2;
    return rect(p0[first], p1[first], p0[second], p1[second], k, axis, m);
}
ShapeBuilder& ShapeBuilder::rectXY(const Vector3& p0, const Vector3& p1, float k, const MaterialPtr& m) {
    return rect(p0, p1, k, Rect::kXY, m);
}
ShapeBuilder& ShapeBuilder::rectXZ(const Vector3& p0, const Vector3& p1, float k, const MaterialPtr& m) {
    return rect(p0, p1, k, Rect::kXZ, m);
}
ShapeBuilder& ShapeBuilder::rectYZ(const Vector3& p0, const Vector3& p1, float k, const MaterialPtr& m) {
    return rect(p0, p1, k, Rect::kYZ, m);
}
ShapeBuilder& ShapeBuilder::box(const Vector3& p0, const Vector3& p1, const MaterialPtr& m) {
    m_ptr = std::make_shared<Box>(p0, p1, m);
    return *this;
}
ShapeBuilder& ShapeBuilder::flip() {
    m_ptr = std::make_shared<FlipNormals>(m_ptr);
    return *this;
}
ShapeBuilder& ShapeBuilder::translate(const Vector3& t) {
    m_ptr = std::make_shared<Translate>(m_ptr, t);
    return *this;
}
ShapeBuilder& ShapeBuilder::rotate(const Vector3& axis, float angle) {
    m_ptr = std::make_shared<Rotate>(m_ptr, axis, angle);
    return *this;
}

// ---------------------------------------------------------------------------
// image
// ---------------------------------------------------------------------------
Image::Image(int w, int h) : m_width(w), m_height(h), m_pixels((size_t)w * (size_t)h) {}

void Image::write(int x, int y, float r, float g, float b) {
    float c[3] = {r, g, b};
    // DenanFilter.h:8-13: a NaN channel becomes 1 / 0 / 1 (magenta)
    const float nan_fill[3] = {1.0f, 0.0f, 1.0f};
    for (int k = 0; k < 3; ++k)
        if (!(c[k] == c[k])) c[k] = nan_fill[k];
    // GammaFilter.h:7-9 -> linear_to_gamma, Util.h:41-47
    const float recipGamma = 1.f / GAMMA_FACTOR;
    for (int k = 0; k < 3; ++k) c[k] = ::powf(c[k], recipGamma);
    // TonemapFilter.h:8-10: max with 0 then min with 1 (a NaN ends at 0)
    for (int k = 0; k < 3; ++k) {
        float m = (c[k] > 0.f) ? c[k] : 0.f;
        c[k] = (m < 1.f) ? m : 1.f;
    }
    rgb& px = m_pixels[(size_t)m_width * y + x];
    px.r = static_cast<unsigned char>(c[0] * 255.99f);
    px.g = static_cast<unsigned char>(c[1] * 255.99f);
    px.b = static_cast<unsigned char>(c[2] * 255.99f);
}

bool Image::save_bmp(const char* path) const {
    FILE* fp = std::fopen(path, "wb");
    if (!fp) return false;
    const int pad = (-m_width * 3) & 3;
    const uint32_t data_bytes = (uint32_t)((m_width * 3 + pad) * m_height);
    unsigned char hdr[54];
    std::memset(hdr, 0, sizeof hdr);
    auto put32 = [&](int at, uint32_t v) {
        hdr[at] = v & 0xff; hdr[at + 1] = (v >> 8) & 0xff; hdr[at + 2] = (v >> 16) & 0xff; hdr[at + 3] = (v >> 24) & 0xff;
    };
    auto put16 = [&](int at, uint32_t v) { hdr[at] = v & 0xff; hdr[at + 1] = (v >> 8) & 0xff; };
    hdr[0] = 'B';
    hdr[1] = 'M';
    put32(2, 54 + data_bytes);  // file size
    put32(10, 54);              // pixel data offset
    put32(14, 40);              // BITMAPINFOHEADER
    put32(18, (uint32_t)m_width);
    put32(22, (uint32_t)m_height);
    put16(26, 1);   // planes
    put16(28, 24);  // bits per pixel
    bool ok = std::fwrite(hdr, 1, sizeof hdr, fp) == sizeof hdr;
    std::vector<unsigned char> row((size_t)m_width * 3 + pad, 0);
    for (int y = m_height - 1; y >= 0 && ok; --y) {  // bottom-up rows, BGR
        const rgb* src = &m_pixels[(size_t)y * m_width];
        for (int x = 0; x < m_width; ++x) {
            row[3 * x + 0] = src[x].b;
            row[3 * x + 1] = src[x].g;
            row[3 * x + 2] = src[x].r;
        }
        ok = std::fwrite(row.data(), 1, row.size(), fp) == row.size();
    }
    ok = (std::fclose(fp) == 0) && ok;
    return ok;
}

// ---------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------
Scene::Scene(const char* fileName, int width, int height, int sample)
    : m_image(new Image(width, height)), m_backColor(0.2f), m_filename(fileName), m_samples(sample) {}

Scene::~Scene() {}

void Scene::build() {
    // The Cornell-box variant the reference hard-codes (Scene.cpp:37-87).
    m_backColor = Vector3(0.0f);
    float aspect = float(m_image->width()) / float(m_image->height());
    m_camera.reset(new Camera(Vector3(278, 278, -800), Vector3(278, 278, 0), Vector3(0, 1, 0), 40, aspect));

    auto solid = [](float r, float g, float b) { return std::make_shared<ColorTexture>(Vector3(r, g, b)); };
    MaterialPtr red = std::make_shared<Lambertian>(solid(0.65f, 0.05f, 0.05f));
    MaterialPtr white = std::make_shared<Lambertian>(solid(0.73f, 0.73f, 0.73f));
    MaterialPtr green = std::make_shared<Lambertian>(solid(0.12f, 0.45f, 0.15f));
    MaterialPtr lamp = std::make_shared<DiffuseLight>(solid(15.0f, 15.0f, 15.0f));
    MaterialPtr aluminum = std::make_shared<Metal>(solid(0.8f, 0.85f, 0.88f), 0.0f);
    MaterialPtr glass = std::make_shared<Dielectric>(1.5f);

    ShapeList* world = new ShapeList();
    ShapeBuilder b;
    world->add(b.rectYZ(0, 555, 0, 555, 555, green).flip().get());
    world->add(b.rectYZ(0, 555, 0, 555, 0, red).get());
    world->add(b.rectXZ(213, 343, 227, 332, 554, lamp).flip().get());
    world->add(b.rectXZ(0, 555, 0, 555, 555, white).flip().get());
    world->add(b.rectXZ(0, 555, 0, 555, 0, white).get());
    world->add(b.rectXY(0, 555, 0, 555, 555, white).flip().get());
    world->add(b.sphere(Vector3(190, 90, 190), 45, aluminum).get());
    world->add(b.sphere(Vector3(380, 90, 100), 90, glass).get());
    world->add(b.box(Vector3(0.0f), Vector3(165, 330, 165), white)
                   .rotate(Vector3::yAxis(), 15)
                   .translate(Vector3(265, 0, 295))
                   .get());
    m_world.reset(world);

    ShapeList* lights = new ShapeList();
    lights->add(b.rectXZ(213, 343, 227, 332, 554, MaterialPtr()).get());
    lights->add(b.sphere(Vector3(190, 90, 190), 90, MaterialPtr()).get());
    m_light.reset(lights);
}

Vector3 Scene::backgroundSky(const Vector3& d) const {
    Vector3 v = normalize(d);
    float t = 0.5f * (v.getY() + 1.0f);
    return lerp(t, Vector3(1.0f), Vector3(0.5f, 0.7f, 1.0f));
}

bool Scene::flatten(FlatScene& out) const {
    if (!m_camera || !m_world) {
        out.fail("scene needs a camera and a world");
        return false;
    }
    FlatScene::Cursor cur;
    m_world->flatten(out, cur);
    out.set_general_lights(m_generalLights);
    if (!m_light) {
        // no light shape at all: diffuse bounces sample the material's own pdf only (an
        // extension; the reference's color() dereferences its light unconditionally)
        out.set_light_is_list(true);
    } else if (m_light->is_list()) {
        out.set_light_is_list(true);
        static_cast<const ShapeList*>(m_light.get())->flatten_as_light_list(out);
        if (out.light_kind.empty())
            out.fail("the light list is empty (undefined behaviour in the reference, ShapeList.cpp:28-35)");
    } else {
        out.set_light_is_list(false);
        out.n_top_lights = 1;
        FlatScene::Cursor lcur;
        m_light->flatten_light(out, out.reserve_lights(1), lcur);
    }
    if (out.prim_kind.empty()) out.fail("the world has no primitives");
    out.set_camera(*m_camera);
    out.set_background(m_backColor);
    out.set_background_sky(m_backgroundSky);
    return out.error().empty();
}

bool Scene::render() {
    build();
    m_error.clear();
    FlatScene flat;
    if (!flatten(flat)) {
        m_error = flat.error();
        return false;
    }
    const int nx = m_image->width(), ny = m_image->height();
    const std::vector<int> devices = m_devices.empty() ? std::vector<int>{m_device} : m_devices;
    KernelCompilation compile = m_kernelCompilation;
    if (compile == KernelCompilation::Auto)
        compile = (unsigned long long)nx * (unsigned long long)ny * (unsigned long long)m_samples >= kCompileFromSamples
                      ? KernelCompilation::AtUpload : KernelCompilation::Never;
    std::vector<RtxContext*> ctxs;
    bool ok = true;
    for (int d : devices) {
        RtxContext* ctx = nullptr;
        ok = rtx_create(d, &ctx) == RTX_OK && (ctxs.push_back(ctx), true) &&
             rtx_set_option(ctx, RTX_OPTION_SPECIALIZE_SCAN, (uint64_t)compile) == RTX_OK &&
             rtx_upload_scene(ctx, flat.view(), RTX_ACCEL_AUTO) == RTX_OK;
        if (!ok) break;
    }
    std::vector<float> accum((size_t)nx * ny * 3, 0.0f);
    m_rays = 0;
    auto t0 = std::chrono::steady_clock::now();
    // a launch is m_samplesPerLaunch samples per pixel on EACH device
    const int per_call = m_samplesPerLaunch * (int)devices.size();
    for (int done = 0; ok && done < m_samples;) {
        RtxParams p;
        std::memset(&p, 0, sizeof p);
        p.width = (uint32_t)nx;
        p.height = (uint32_t)ny;
        p.spp_begin = (uint32_t)done;
        p.spp_count = (uint32_t)std::min(per_call, m_samples - done);
        p.max_depth = (uint32_t)m_maxDepth;
        p.mode = m_deterministic ? RTX_MODE_DETERMINISTIC : RTX_MODE_FAST;
        p.seed = m_seed;
        p.accumulate = done > 0 ? 1u : 0u;
        uint64_t rays = 0;
        ok = (ctxs.size() == 1 ? rtx_render(ctxs[0], &p, accum.data(), &rays)
                               : rtx_render_multi(ctxs.data(), (uint32_t)ctxs.size(), &p, accum.data(), &rays)) == RTX_OK;
        m_rays += rays;
        done += (int)p.spp_count;
    }
    m_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!ok) m_error = rtx_last_error();
    for (RtxContext* ctx : ctxs) rtx_destroy(ctx);
    if (!ok) return false;
    // Scene.cpp:150-151: c /= ns, then the filter chain; accumulator rows are already
    // top-first.
    const float ns = (float)m_samples;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            const float* c = &accum[((size_t)y * nx + x) * 3];
            m_image->write(x, y, c[0] / ns, c[1] / ns, c[2] / ns);
        }
    if (!m_image->save_bmp(m_filename.c_str())) {
        m_error = "cannot write " + m_filename;
        return false;
    }
    return true;
}

}  // namespace rt
