/*
 * rtx.h — C ABI of the MI355X path-trace library (librtx.so).
 *
 * The reference tracer has no plugin / FFI boundary: main() calls
 * Scene::render() directly (Raytrace_C++/Src/main.cpp:11-12) and render() runs
 * the pixel / sample / path loops on the host (Scene.cpp:134-156).  This header
 * is the boundary placed where Scene::render() hands off to per-pixel work
 * (Scene.cpp:137-153): the host scene API flattens world / light / camera into
 * the plain arrays of RtxScene and the library renders them on the GPU.
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer; the
 * library copies what it needs during the call and never keeps a host pointer;
 * every entry point returns 0 on success or a negative RtxStatus and never
 * throws; rtx_last_error() describes the last failure on the calling thread.
 * One context drives one GPU; a context must not be used from two threads at
 * once, different contexts may be.
 */
#ifndef RTX_H
#define RTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 5u

typedef enum RtxStatus {
    RTX_OK = 0,
    RTX_ERR_INVALID_ARGUMENT = -1,
    RTX_ERR_NO_DEVICE = -2,
    RTX_ERR_HIP = -3,
    RTX_ERR_NO_SCENE = -4,
    RTX_ERR_UNSUPPORTED = -5,
    RTX_ERR_TOO_LARGE = -6
} RtxStatus;

/* Primitive kinds (Rect::AxisType, Rect.h:7-11, plus Sphere). */
enum { RTX_PRIM_RECT_XY = 0, RTX_PRIM_RECT_XZ = 1, RTX_PRIM_RECT_YZ = 2, RTX_PRIM_SPHERE = 3 };
/* Light entry kinds; OTHER = any shape that keeps Shape's default pdf_value()/random()
 * (Shape.h:8-9: 0 and (1,0,0)); LIST = a ShapeList nested inside the light shape
 * (ShapeList.cpp:19-35 recurse into it). */
enum {
    RTX_LIGHT_RECT_XY = 0, RTX_LIGHT_RECT_XZ = 1, RTX_LIGHT_RECT_YZ = 2, RTX_LIGHT_SPHERE = 3, RTX_LIGHT_OTHER = 4,
    RTX_LIGHT_LIST = 5
};
/* Which shapes can be sampled as lights.
 *   REFERENCE  exactly the reference: only XZ rects (Rect.cpp:37,51) and spheres have a density
 *              and a sampler; XY / YZ rects and every FlipNormals / Translate / Rotate / Box
 *              wrapper keep Shape's defaults, 0 and (1,0,0) (Shape.h:8-9).
 *   GENERAL    an extension with no counterpart in the reference: rects of all three axes are
 *              sampled by area like the XZ one, and a leaf entry may sit under a transform
 *              chain: pdf_value(o, v) and random(o) take o (and v) into the leaf's frame the way
 *              Translate::hit / Rotate::hit take a ray (Translate.cpp:7, Rotate.cpp:7-9), and
 *              random()'s direction is rotated back out (Rotate.cpp:13). */
enum { RTX_LIGHTS_REFERENCE = 0, RTX_LIGHTS_GENERAL = 1 };
/* Transform steps (Translate.h:7, Rotate.h:7). */
enum { RTX_STEP_TRANSLATE = 0, RTX_STEP_ROTATE = 1 };
/* Material kinds (Lambertian.h, Metal.h, Dielectric.h, DiffuseLight.h). */
enum { RTX_MAT_LAMBERTIAN = 0, RTX_MAT_METAL = 1, RTX_MAT_DIELECTRIC = 2, RTX_MAT_DIFFUSE_LIGHT = 3 };
/* Texture kinds (ColorTexture.h, CheckerTexture.h, ImageTexture.h).  SKY has no class of its own in
 * the reference: its value at a point p is Scene::backgroundSky(p - centre) (Scene.h:24-28), the
 * gradient the reference defines for rays that hit nothing; on an emitting sphere around the
 * scene it is the sky as a surface, which the reference's own Scene::color can render. */
enum { RTX_TEX_COLOR = 0, RTX_TEX_CHECKER = 1, RTX_TEX_IMAGE = 2, RTX_TEX_SKY = 3 };
/* What a ray that hits nothing returns (Scene::background / Scene::backgroundSky, Scene.h:20-28). */
enum { RTX_BACKGROUND_CONSTANT = 0, RTX_BACKGROUND_SKY = 1 };
/* Acceleration structure choice for rtx_upload_scene(). */
enum { RTX_ACCEL_AUTO = 0, RTX_ACCEL_LINEAR = 1, RTX_ACCEL_BVH = 2 };
/* Arithmetic modes of a render. */
enum {
    RTX_MODE_FAST = 0,          /* fused multiply-add, reciprocal-based division, native sin/cos */
    RTX_MODE_DETERMINISTIC = 1  /* the reference's exact fp32 operation order; path geometry is
                                   bit-identical to the reference tracer fed the same stream */
};

/*
 * A flattened scene: struct of arrays, host memory, read-only during the call.
 *
 * Primitives are the leaves (Rect / Sphere) of the world shape in depth-first
 * order, which is the order ShapeList::hit (ShapeList.cpp:5-17) visits them;
 * Box (Box.cpp:11-29) contributes its six rects.  FlipNormals wrappers collapse
 * to a parity bit; Translate / Rotate wrappers become an ordered chain of steps
 * (outermost first) shared by every leaf below them.
 */
typedef struct RtxScene {
    uint32_t abi_version; /* RTX_ABI_VERSION */

    /* world primitives */
    uint32_t n_prims;
    const uint32_t* prim_kind;     /* RTX_PRIM_* */
    const uint32_t* prim_flip;     /* 1 = normal negated (FlipNormals.cpp:8) */
    const uint32_t* prim_material; /* index into mat_* */
    const uint32_t* prim_chain;    /* index into chain_begin; chain 0 is the empty chain */
    const float* prim_a;           /* rect x0 (Rect.h:30) | sphere centre x */
    const float* prim_b;           /* rect x1             | sphere centre y */
    const float* prim_c;           /* rect y0             | sphere centre z */
    const float* prim_d;           /* rect y1             | sphere radius   */
    const float* prim_e;           /* rect k              | unused          */

    /* transform chains: steps chain_begin[c] .. chain_begin[c+1]-1, outermost wrapper first */
    uint32_t n_chains;             /* >= 1; chain 0 has no steps */
    const uint32_t* chain_begin;   /* n_chains + 1 entries */
    uint32_t n_steps;
    const uint32_t* step_kind;     /* RTX_STEP_* */
    const float* step_x;           /* translate: offset xyz, w unused (Translate.cpp:7-9)      */
    const float* step_y;           /* rotate: unit quaternion xyzw (Rotate.h:9, quat_aos.h:348) */
    const float* step_z;
    const float* step_w;

    /* materials */
    uint32_t n_materials;
    const uint32_t* mat_kind;      /* RTX_MAT_* */
    const uint32_t* mat_texture;   /* albedo / emission texture; unused for dielectrics */
    const float* mat_param;        /* Metal: fuzz (Metal.h:23); Dielectric: ri (Dielectric.h:14) */

    /* textures */
    uint32_t n_textures;
    const uint32_t* tex_kind;      /* RTX_TEX_* */
    const float* tex_r;            /* colour: rgb (ColorTexture.h:15); sky: the centre xyz  */
    const float* tex_g;            /* checker: tex_r = frequency (CheckerTexture.h:18)    */
    const float* tex_b;
    const uint32_t* tex_odd;       /* checker: sub-texture when the sine product is < 0; image: width  */
    const uint32_t* tex_even;      /* checker: sub-texture otherwise (CheckerTexture.cpp:5-10); image: height */
    const uint32_t* tex_offset;    /* image: byte offset of its first texel in `texels` (ImageTexture.h:26) */
    uint64_t n_texel_bytes;        /* may be 0 */
    const uint8_t* texels;         /* 8-bit RGB, rows top-down, 3 bytes per texel, no padding */

    /* light shape handed to Scene::color (Scene.cpp:113).  n_lights == 0 (an extension: the
     * reference's color() always takes a light) means there is none: a diffuse bounce is then
     * sampled from the material's own pdf alone, srec.pdf of Lambertian.cpp:19, with no
     * mixture and no extra draw. */
    uint32_t n_lights;             /* entries of the top-level list, or 1 for a bare shape */
    uint32_t light_is_list;        /* 1: a ShapeList (one extra draw picks the entry,
                                      ShapeList.cpp:28-35); 0: a single bare shape */
    const uint32_t* light_kind;    /* RTX_LIGHT_*; n_light_entries of each light_* array */
    const float* light_a;          /* same meaning as prim_a..prim_e */
    const float* light_b;
    const float* light_c;
    const float* light_d;
    const float* light_e;

    /* pinhole camera (Camera.h:19-26): ray = origin, ll + u*s + v*t - origin */
    float cam_origin[3];
    float cam_ll[3];
    float cam_u[3];
    float cam_v[3];

    float background[3];           /* Scene::background (Scene.h:20-22) */
    uint32_t background_mode;      /* RTX_BACKGROUND_*; SKY ignores background[] */

    /* Light trees.  Entries [0, n_lights) of the light_* arrays are the top-level list (or the
     * one bare shape).  A ShapeList nested in it is an entry of kind RTX_LIGHT_LIST; its
     * children are the entries [light_first[l], light_first[l] + light_count[l]), which lie
     * behind the top-level ones and may be lists again.  A nested list draws one more number to
     * pick a child and weighs its children's densities by 1 / count (ShapeList.cpp:19-35). */
    uint32_t n_light_entries;      /* length of the light_* arrays; 0 means n_lights */
    uint32_t light_sampling;       /* RTX_LIGHTS_* */
    const uint32_t* light_first;   /* LIST entries only; may be NULL when there is none */
    const uint32_t* light_count;   /* LIST entries only, >= 1 (the reference indexes an empty
                                      list out of bounds, ShapeList.cpp:28-35) */
    const uint32_t* light_chain;   /* RTX_LIGHTS_GENERAL: index into chain_begin of each leaf
                                      entry's transform chain; NULL = none has one */
} RtxScene;

typedef struct RtxParams {
    uint32_t width;
    uint32_t height;
    uint32_t spp_begin;   /* absolute index of the first sample rendered by this call */
    uint32_t spp_count;   /* samples per pixel rendered by this call */
    uint32_t max_depth;   /* MAX_DEPTH (Scene.cpp:9): scatter is skipped at this depth; 50 */
    uint32_t mode;        /* RTX_MODE_* */
    uint64_t seed;
    uint32_t accumulate;  /* 0: overwrite the accumulator; 1: add to it */
    uint32_t reserved;
} RtxParams;

/* One nearest-hit record of rtx_hit(), the image of the reference's HitRec (HitRec.h:3-10). */
typedef struct RtxHit {
    int32_t hit;          /* 1 / 0 */
    int32_t prim;         /* index of the winning primitive, -1 on a miss */
    float t;
    float p[3];
    float n[3];
    int32_t material;     /* material index, -1 on a miss */
    float uv[2];          /* texture coordinates (Rect.cpp:27-28, Util.h:78-83) */
} RtxHit;

typedef struct RtxContext RtxContext;

/* Library / device bring-up.  `device` is a HIP device ordinal. */
int rtx_create(int device, RtxContext** out_ctx);
void rtx_destroy(RtxContext* ctx);
const char* rtx_last_error(void);
/* Number of HIP devices visible to this process; 0 when there is none or the runtime cannot be
 * initialised (never fails). */
int rtx_device_count(void);

/*
 * Options of a context, set before the calls they affect.  Returns RTX_ERR_INVALID_ARGUMENT for an
 * unknown option or value.
 *   RTX_OPTION_SPECIALIZE_SCAN  (default 1) 1: rtx_upload_scene compiles the fast-mode kernels of a small
 *       scene (no tree, at most 64 scan records: the reference's own scene, Scene.cpp:43-86, has
 *       14) for that scene with the HIP run-time compiler: its primitives become literals of a
 *       straight-line scan instead of a table walked at run time, so do the entries of its light
 *       list (ShapeList.cpp:19-26), and tests for what the scene does not hold (textured surfaces,
 *       fuzzy metals, a kind of light) are left out.  The same tests in the same order, so the same
 *       primitive wins a ray but for candidates within rounding of each other (the compiler may
 *       contract a product with a literal differently from one with a register); frames, per-sample
 *       outputs and rtx_hit of a scene all come from the compiled scan.  An upload takes 0.2 - 0.7 s
 *       longer the first time a process sees the scene.  Where the run-time compiler is missing or
 *       fails the table-driven kernels render (rtx_scene_info and rtx_scene_note say so).  0: never.
 *       2: in the background - rtx_upload_scene hands the compilation to a helper program (rtx_compile,
 *       installed beside the library; a thread of the library waits for it) and returns at once; the
 *       table-driven kernels render until the compiled ones are ready, and the first render call (or
 *       rtx_scene_info) after that takes them over, which loads a code object on the device
 *       (rtx_render_noalloc never does: it goes on with what the context has).  rtx_destroy and a new
 *       upload end a compilation that is still running; a process may exit while one runs.  Samples
 *       rendered before and after the switch come from two compilations of the same estimator and
 *       differ in rounding only; which call switches depends on timing, so choose 0 or 1 where two runs
 *       must give the same bits.  Without the helper program the table-driven kernels stay
 *       (rtx_scene_note says so).  When does a compilation pay: it costs about as much as 3 s of
 *       rendering gain back (x 1.07 - 1.22, more with dozens of lights).
 *       Deterministic mode is not affected.
 */
enum { RTX_OPTION_SPECIALIZE_SCAN = 1 };
int rtx_set_option(RtxContext* ctx, uint32_t option, uint64_t value);

/*
 * Compile the scene's specialised kernel ahead of its upload, on the calling thread; needs no
 * context and no GPU (an application can do it on a worker thread while it loads its assets: the
 * upload of the same scene then finds the code object in the process's cache).  RTX_OK when the
 * code object is in the cache afterwards; RTX_ERR_UNSUPPORTED when the scene is not one whose scan
 * is specialised or the run-time compiler is missing or failed (rtx_last_error says which).
 * A thread that is inside this call must be joined before the process exits: the run-time compiler
 * does not survive the teardown of its own static objects (value 2 of RTX_OPTION_SPECIALIZE_SCAN has no such
 * condition, it compiles in a helper process).
 */
int rtx_prepare_scene(const RtxScene* scene);
/*
 * The generated part of that kernel's source, for inspection: the scene's feature set
 * (RTX_SPECIAL_FEATURES), its scan as the body of the RTX_SPECIAL_SCAN macro (one line per transform
 * group and per record, in the order the table-driven kernel tests them, operands as the bits of
 * the fp32 values) and, for a light list without a tree, its entries (RTX_SPECIAL_LIGHTS); the rest of the
 * translation unit is raytrace_amd/csrc/rtx_core.h as built into the library.  Copies at most
 * `capacity` bytes (NUL-terminated) into `text` and stores the size of the whole text, NUL
 * included, in *needed (either may be null).  Errors as rtx_prepare_scene.
 */
int rtx_scene_kernel_source(const RtxScene* scene, char* text, uint64_t capacity, uint64_t* needed);

/* Copy the scene to the GPU (and build the BVH when asked or when AUTO decides). */
int rtx_upload_scene(RtxContext* ctx, const RtxScene* scene, uint32_t accel);

/*
 * Render samples [spp_begin, spp_begin+spp_count) of every pixel: the device
 * counterpart of the loops of Scene::render (Scene.cpp:140-149) around
 * Scene::color (Scene.cpp:103-132).
 *   accum_rgb   HOST buffer, width*height*3 fp32, row 0 = top image row (the
 *               reference writes pixel (i,j) to row ny-j-1, Scene.cpp:151); receives
 *               the SUM of the samples' radiance (divide by the sample count to get
 *               the reference's `c /= ns`).
 *   rays_out    optional: nearest-hit queries issued (one per path segment).
 */
int rtx_render(RtxContext* ctx, const RtxParams* params, float* accum_rgb, uint64_t* rays_out);

/*
 * rtx_render without any allocation.  No counterpart in the reference: it is for callers that
 * cannot allocate device memory at the time of the call (tight serving loops, harnesses that
 * forbid hipMalloc; the call itself still copies and synchronises, so it is not for use inside a
 * stream capture).  The device-side accumulator and ray counter,
 * which rtx_render keeps in buffers its context allocates on first use, live in one file-scope
 * device array of the library instead (RTX_NOALLOC_MAX_PIXELS * 12 + 256 bytes, held by every
 * GPU the process has created a context on until the process ends); the call makes no hipMalloc,
 * hipFree or other allocating call.  Arguments and result as for rtx_render (bit-identical to
 * it).  Frames of more than RTX_NOALLOC_MAX_PIXELS pixels are refused with RTX_ERR_TOO_LARGE.
 * Every context on one GPU shares that GPU's array: calls that overlap on one GPU race.
 */
#define RTX_NOALLOC_MAX_PIXELS (1u << 21) /* 2 097 152: a 1920x1080 frame fits */
int rtx_render_noalloc(RtxContext* ctx, const RtxParams* params, float* accum_rgb, uint64_t* rays_out);

/*
 * The same frame on several GPUs from ONE process: the sample range
 * [spp_begin, spp_begin+spp_count) is cut into n_ctx contiguous blocks (the first
 * spp_count % n_ctx blocks one sample longer), block i is rendered by ctxs[i] on that
 * context's device, all devices at the same time; the partial sums are copied
 * device-to-device to ctxs[0]'s GPU (hipMemcpyPeerAsync, one copy per context over its own
 * link; no RCCL in this library: a 1080p frame is 24.9 MB per GPU, and a process that wants
 * the ring reduce runs one rank per GPU over torch.distributed instead) and added there in
 * context order, so the result does not depend on timing.  Samples are keyed by their absolute index: up to the order of the
 * fp32 additions the frame is the one rtx_render makes on one GPU.  Every context must
 * hold the same scene (rtx_upload_scene on each); two contexts may share a device.
 * The reference has one loop nest on one host (Scene.cpp:137-153; its OpenMP pragma over
 * pixels, Scene.cpp:139, is the nearest thing to a work split); this entry point is for C
 * and C++ callers - Python callers use one process per GPU and torch.distributed instead.
 *   accum_rgb / rays_out   as for rtx_render.
 */
int rtx_render_multi(RtxContext* const* ctxs, uint32_t n_ctx, const RtxParams* params, float* accum_rgb,
                     uint64_t* rays_out);

/*
 * Same, but the accumulator and the ray counter are DEVICE buffers on the
 * context's GPU and the work is enqueued on `hip_stream` (a hipStream_t, NULL =
 * the default stream) without waiting for it.  `d_rays` (optional, 8 bytes) is
 * incremented, not overwritten.
 */
int rtx_render_device(RtxContext* ctx, const RtxParams* params, float* d_accum_rgb,
                      unsigned long long* d_rays, void* hip_stream);

/*
 * Deterministic-mode debugging aid: per-sample radiance and segment counts.
 *   sample_rgb       HOST, [height*width][spp_count][3], pixel index j*width+i with j
 *                    counted from the BOTTOM row (the reference's loop variables).
 *   sample_segments  HOST, optional, [height*width][spp_count].
 */
int rtx_render_samples(RtxContext* ctx, const RtxParams* params, float* sample_rgb,
                       uint32_t* sample_segments);

/*
 * Nearest hit of n arbitrary rays in (t0, t1): Shape::hit of the world
 * (Shape.h:7, ShapeList.cpp:5-17).  `origins` / `directions` are HOST arrays of
 * 3n floats; `hits` is a HOST array of n records.  mode as in RtxParams.
 */
int rtx_hit(RtxContext* ctx, uint32_t n, const float* origins, const float* directions, float t0,
            float t1, uint32_t mode, RtxHit* hits);

/*
 * The reference's per-pixel epilogue (Scene.cpp:150-151, Image.h:30-39):
 * sum / ns, NaN filter, gamma 1/2.2, clamp, *255.99 truncated.  `d_accum_rgb` is a
 * DEVICE accumulator, `rgb8` a HOST buffer of width*height*3 bytes, row 0 = top.
 */
int rtx_resolve_device(RtxContext* ctx, uint32_t width, uint32_t height, uint32_t ns,
                       const float* d_accum_rgb, uint8_t* rgb8, void* hip_stream);

/*
 * The same epilogue (Scene.cpp:150-151, Image.h:30-39) with the 8-bit image left on the GPU:
 * `d_rgb8` is a DEVICE buffer of width*height*3 bytes.  Only enqueues the kernel on
 * `hip_stream` and returns: for callers that display or encode the frame on the GPU, or that copy
 * it back on a stream of their own while the next frame renders.
 */
int rtx_resolve_to_device(RtxContext* ctx, uint32_t width, uint32_t height, uint32_t ns,
                          const float* d_accum_rgb, uint8_t* d_rgb8, void* hip_stream);

/* Introspection for tests and reports. */
typedef struct RtxSceneInfo {
    uint32_t n_prims;
    uint32_t n_items;      /* top-level BVH items (single primitives or whole transformed instances) */
    uint32_t n_bvh_nodes;  /* 0 when the linear scan is used */
    uint32_t accel;        /* RTX_ACCEL_LINEAR or RTX_ACCEL_BVH actually in use */
    uint32_t bvh_depth;    /* inner-node levels above the deepest leaf; 0 without a tree */
    float bvh_visits;      /* the builder's surface-area estimate of the inner nodes a ray through the
                              root box visits (sum of the inner nodes' areas / the root's area) */
    uint32_t scan_specialized; /* 1: fast-mode frames of this scene run the kernel compiled for it at upload */
    float specialize_seconds;  /* what that compilation took in this upload (0: reused, or none) */
} RtxSceneInfo;
int rtx_scene_info(RtxContext* ctx, RtxSceneInfo* out);
/* Why the scene's scan was not specialised ("" when it was, or when the scene is not a candidate);
 * valid until the next upload on the context. */
const char* rtx_scene_note(RtxContext* ctx);

#ifdef __cplusplus
}
#endif

#endif /* RTX_H */
