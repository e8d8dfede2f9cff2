// rtx_special.cpp — kernels compiled for one scene: source generation (scan, light list, feature set),
// run-time compilation, the process's cache of code objects, compilations on a thread of their own.
#include "rtx_special.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>

extern "C" {
extern const char rtx_src_core[];    // rtx_core.h    (rtx_sources.cpp)
extern const char rtx_src_device[];  // rtx_device.h
extern const char rtx_src_det[];     // include/rtx_det.h
extern const char rtx_src_abi[];     // include/rtx.h
}

namespace rtxspecial {
namespace {

uint32_t bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, sizeof u);
    return u;
}

void append(std::string& s, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void append(std::string& s, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    s += buf;
}

// The HIP run-time compiler, taken from the process at first use (no link-time dependency: a
// process without it renders through the table-driven kernels).
struct Rtc {
    using Program = void*;
    int (*create)(Program*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*add_name)(Program, const char*) = nullptr;
    int (*compile)(Program, int, const char* const*) = nullptr;
    int (*log_size)(Program, size_t*) = nullptr;
    int (*log)(Program, char*) = nullptr;
    int (*lowered)(Program, const char*, const char**) = nullptr;
    int (*code_size)(Program, size_t*) = nullptr;
    int (*code)(Program, char*) = nullptr;
    int (*destroy)(Program*) = nullptr;
    bool ok = false;
    std::string why;
};

// (this and the cache below are never destroyed: threads of the application may still compile or load while
// the process's static objects go)
const Rtc& rtc() {
    static Rtc& r = *new Rtc;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"libhiprtc.so.7", "libhiprtc.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) {
            r.why = "the HIP run-time compiler (libhiprtc.so) is not available";
            return;
        }
        auto sym = [&](const char* n) { return dlsym(h, n); };
        r.create = reinterpret_cast<decltype(r.create)>(sym("hiprtcCreateProgram"));
        r.add_name = reinterpret_cast<decltype(r.add_name)>(sym("hiprtcAddNameExpression"));
        r.compile = reinterpret_cast<decltype(r.compile)>(sym("hiprtcCompileProgram"));
        r.log_size = reinterpret_cast<decltype(r.log_size)>(sym("hiprtcGetProgramLogSize"));
        r.log = reinterpret_cast<decltype(r.log)>(sym("hiprtcGetProgramLog"));
        r.lowered = reinterpret_cast<decltype(r.lowered)>(sym("hiprtcGetLoweredName"));
        r.code_size = reinterpret_cast<decltype(r.code_size)>(sym("hiprtcGetCodeSize"));
        r.code = reinterpret_cast<decltype(r.code)>(sym("hiprtcGetCode"));
        r.destroy = reinterpret_cast<decltype(r.destroy)>(sym("hiprtcDestroyProgram"));
        r.ok = r.create && r.add_name && r.compile && r.log_size && r.log && r.lowered && r.code_size && r.code && r.destroy;
        if (!r.ok) r.why = "libhiprtc.so lacks an entry point";
    });
    return r;
}

// The kernels compiled around a scan: the fast-mode trace kernel of the scene's variant for frames and
// for per-sample output, and the ray query.
constexpr int kKernels = 3;
std::string kernel_expr(int k, Variant v) {
    if (k == 2) return "rtx_fast_mode::hit_kernel<false>";  // rtx_hit
    return std::string("rtx_fast_mode::trace_kernel<false, ") + (k == 1 ? "true" : "false") + ", " + (v.light_tree ? "true" : "false") + ", " +
           (v.list_lit ? "true" : "false") + ">";
}

struct CodeObject {
    std::vector<char> image;
    std::string kernel[kKernels];  // lowered names
};
std::mutex& g_cache_mutex = *new std::mutex;
std::map<std::string, std::shared_ptr<CodeObject>>& g_cache = *new std::map<std::string, std::shared_ptr<CodeObject>>;  // by variant and scan text
// (a process that sees ever new scenes keeps the kMaxCached most recently compiled ones: a code object
// is about 100 KB, a loaded module keeps its own copy)
constexpr size_t kMaxCached = 256;
std::deque<std::string>& g_cache_order = *new std::deque<std::string>;

bool compile(const std::string& scan, Variant variant, CodeObject& out, std::string& why) {
    const Rtc& r = rtc();
    if (!r.ok) {
        why = r.why;
        return false;
    }
    // the translation unit: rtx_kernels_fast.hip's two lines with the scan in front
    std::string src = "#define RTX_DET 0\n#define RTX_TRACE_WAVES 8\n";
#ifdef RTX_VARIANT_DEFINES  // (experiment builds, tools/build_variant.sh: the defines of the variant's own fast-mode kernels)
    src += RTX_VARIANT_DEFINES;
#endif
    src += scan + "#include \"rtx_core.h\"\n";
    const char* headers[] = {rtx_src_core, rtx_src_device, rtx_src_det, rtx_src_abi};
    const char* names[] = {"rtx_core.h", "rtx_device.h", "rtx_det.h", "rtx.h"};
    Rtc::Program prog = nullptr;
    if (r.create(&prog, src.c_str(), "rtx_special.hip", 4, headers, names) != 0) {
        why = "hiprtcCreateProgram failed";
        return false;
    }
    bool ok = false;
    do {
        bool named = true;
        for (int k = 0; k < kKernels; ++k) named = named && r.add_name(prog, kernel_expr(k, variant).c_str()) == 0;
        if (!named) {
            why = "hiprtcAddNameExpression failed";
            break;
        }
        // the flags of the fast-mode translation unit (csrc/Makefile: COMMON + FASTFLAGS)
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=fast",
                              "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fno-math-errno", "-fno-signed-zeros",
                              "-fno-trapping-math", "-Wno-unused-function"};
        if (r.compile(prog, (int)(sizeof opts / sizeof opts[0]), opts) != 0) {
            size_t n = 0;
            r.log_size(prog, &n);
            std::string log(n, '\0');
            if (n) r.log(prog, &log[0]);
            why = "run-time compilation failed: " + log.substr(0, 600);
            break;
        }
        bool lowered_all = true;
        for (int k = 0; k < kKernels; ++k) {
            const char* lowered = nullptr;
            if (r.lowered(prog, kernel_expr(k, variant).c_str(), &lowered) != 0 || !lowered) {
                lowered_all = false;
                break;
            }
            out.kernel[k] = lowered;
        }
        if (!lowered_all) {
            why = "hiprtcGetLoweredName failed";
            break;
        }
        size_t n = 0;
        if (r.code_size(prog, &n) != 0 || n == 0) {
            why = "hiprtcGetCodeSize failed";
            break;
        }
        out.image.resize(n);
        if (r.code(prog, out.image.data()) != 0) {
            why = "hiprtcGetCode failed";
            break;
        }
        ok = true;
    } while (false);
    r.destroy(&prog);
    return ok;
}

}  // namespace

std::string lights_source(const std::vector<rtxd::Light>& lights, uint32_t n_lights, bool is_list) {
    if (n_lights == 0u || n_lights > kMaxRecords) return std::string();
    std::string s = "#define RTX_SPECIAL_LIGHTS \\\n";
    for (uint32_t l = 0; l < n_lights; ++l) {
        const rtxd::Light& E = lights[l];
        append(s, "    RTX_SPECIAL_LIGHT(%uu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu); \\\n", E.kind, bits(E.a), bits(E.b), bits(E.c),
               bits(E.d), bits(E.e), bits(E.inv_area));
    }
    // ShapeList::pdf_value weighs each entry by 1 / size (ShapeList.cpp:20); a bare shape counts whole
    append(s, "\n#define RTX_SPECIAL_LIGHT_WEIGHT 0x%08xu\n", bits(is_list ? 1.0f / (float)n_lights : 1.0f));
    return s;
}

std::string features_source(const std::vector<rtxd::ScanHit>& hits, const std::vector<rtxd::Light>& lights, uint32_t n_lights,
                            bool light_tree) {
    // (the bit values are those of kFeat* in rtx_core.h)
    uint32_t f = 0u;
    for (const rtxd::ScanHit& H : hits) {
        if (H.bits & rtxd::kInfoTextured) f |= 1u;
        if (H.bits & rtxd::kInfoNeedsUV) f |= 2u;
        switch ((H.bits >> rtxd::kInfoMatShift) & 3u) {
            case RTX_MAT_METAL: f |= 4u | (H.param != 0.0f ? 8u : 0u); break;
            case RTX_MAT_DIELECTRIC: f |= 16u; break;
            case RTX_MAT_LAMBERTIAN: f |= 32u; break;
            default: f |= 64u; break;
        }
    }
    if (light_tree) {
        f |= 128u | 256u | 512u;  // the tree kernels sample through code of their own
    } else {
        for (uint32_t l = 0; l < n_lights; ++l)
            f |= lights[l].kind == RTX_LIGHT_RECT_XZ ? 128u : lights[l].kind == RTX_LIGHT_SPHERE ? 256u : 512u;
    }
    std::string s;
    append(s, "#define RTX_SPECIAL_FEATURES 0x%xu\n", f);
    return s;
}

std::string scan_source(const std::vector<rtxd::ScanGroup>& groups, const std::vector<rtxd::ScanRec>& recs) {
    std::string s = "#define RTX_SPECIAL_SCAN \\\n";
    for (const rtxd::ScanGroup& G : groups) {
        s += "    { \\\n";
        const float* m = G.m;
        if (G.identity) {
            s += "        RTX_SPECIAL_GROUP_IDENTITY(); \\\n";
        } else if (G.rot_y) {
            append(s, "        RTX_SPECIAL_GROUP_ROT_Y(0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu); \\\n", bits(m[0]), bits(m[2]),
                   bits(m[6]), bits(m[8]), bits(m[9]), bits(m[10]), bits(m[11]));
        } else {
            s += "        RTX_SPECIAL_GROUP_AFFINE(";
            for (int k = 0; k < 12; ++k) append(s, "0x%08xu%s", bits(m[k]), k < 11 ? ", " : "); \\\n");
        }
        uint32_t pos = G.first;
        const struct { uint32_t n; const char* axes; } rects[3] = {{G.n_xy, "0, 1, 2"}, {G.n_xz, "0, 2, 1"}, {G.n_yz, "1, 2, 0"}};
        for (const auto& kind : rects)
            for (uint32_t k = 0; k < kind.n; ++k, ++pos) {
                const rtxd::ScanRec& R = recs[pos];
                // (an interval that starts at +0 and ends above it is one unsigned comparison)
                const char* xr = bits(R.a) == 0u && R.b > 0.0f ? "RTX_SPECIAL_RANGE0" : "RTX_SPECIAL_RANGE";
                const char* yr = bits(R.c) == 0u && R.d > 0.0f ? "RTX_SPECIAL_RANGE0" : "RTX_SPECIAL_RANGE";
                append(s, "        RTX_SPECIAL_RECT(%s, %s, %s, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, %uu); \\\n", kind.axes, xr, yr,
                       bits(R.a), bits(R.b), bits(R.c), bits(R.d), bits(R.e), pos);
            }
        if (G.n_sph) s += "        RTX_SPECIAL_SPHERES(); \\\n";
        for (uint32_t k = 0; k < G.n_sph; ++k, ++pos) {
            const rtxd::ScanRec& R = recs[pos];
            append(s, "        RTX_SPECIAL_SPHERE(0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, %uu); \\\n", bits(R.a), bits(R.b), bits(R.c), bits(R.e), pos);
        }
        if (G.n_box) s += "        RTX_SPECIAL_BOXES(); \\\n";
        for (uint32_t k = 0; k < G.n_box; ++k, pos += rtxd::kBoxSlots) {
            const rtxd::ScanRec& R = recs[pos];
            append(s, "        RTX_SPECIAL_BOX(0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, %uu); \\\n", bits(R.a), bits(R.b), bits(R.c),
                   bits(R.d), bits(R.e), bits(R.f), pos);
        }
        s += "    } \\\n";
    }
    s += "\n";
    return s;
}

namespace {
std::string cache_key(const std::string& text, Variant variant) {
    return std::string(variant.light_tree ? "T" : "t") + (variant.list_lit ? "L" : "l") + text;
}

std::shared_ptr<CodeObject> insert_code(const std::string& key, const std::shared_ptr<CodeObject>& fresh) {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    const auto inserted = g_cache.emplace(key, fresh);
    if (inserted.second) {
        g_cache_order.push_back(key);
        while (g_cache_order.size() > kMaxCached) {
            g_cache.erase(g_cache_order.front());
            g_cache_order.pop_front();
        }
    }
    return inserted.first->second;
}

std::shared_ptr<CodeObject> cached_or_compiled(const std::string& scan, Variant variant, std::string& why, double* compile_seconds) {
    const std::string key = cache_key(scan, variant);
    if (compile_seconds) *compile_seconds = 0.0;
    std::shared_ptr<CodeObject> code;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) code = it->second;
    }
    if (!code) {
        auto fresh = std::make_shared<CodeObject>();
        const auto t0 = std::chrono::steady_clock::now();
        if (!compile(scan, variant, *fresh, why)) return nullptr;
        if (compile_seconds) *compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        code = insert_code(key, fresh);
    }
    return code;
}
}  // namespace

bool prepare(const std::string& scan, Variant variant, std::string& why, double* compile_seconds) {
    return cached_or_compiled(scan, variant, why, compile_seconds) != nullptr;
}

bool load(const std::string& scan, Variant variant, Kernel& out, std::string& why, double* compile_seconds) {
    const std::shared_ptr<CodeObject> code = cached_or_compiled(scan, variant, why, compile_seconds);
    if (!code) return false;
    hipModule_t module = nullptr;
    hipError_t e = hipModuleLoadData(&module, code->image.data());
    if (e != hipSuccess) {
        why = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return false;
    }
    hipFunction_t fn[kKernels] = {};
    for (int k = 0; k < kKernels; ++k) {
        e = hipModuleGetFunction(&fn[k], module, code->kernel[k].c_str());
        if (e != hipSuccess) {
            why = std::string("hipModuleGetFunction: ") + hipGetErrorString(e);
            (void)hipGetLastError();
            (void)hipModuleUnload(module);
            return false;
        }
    }
    out.module = module;
    out.frames = fn[0];
    out.samples = fn[1];
    out.hit = fn[2];
    return true;
}

// ---------------------------------------------------------------------------------------------
// Compilations in the background run in a process of their own (the helper rtx_compile beside the
// library): the thread below only feeds it and waits.  A compiler running on a thread of THIS process
// would still be at work when a process that never destroyed its context exits, while the compiler's
// own static objects are torn down under it (measured: an abort, a segmentation fault or a thread that
// never ends); a helper is simply left behind and ends by itself.
// Wire format: to the helper "<light_tree> <list_lit>\n" + the text; back "OK\n", then per kernel
// "<lowered name>\n", then "<bytes>\n" and the code object - or "ERR\n" and the reason.
// ---------------------------------------------------------------------------------------------
bool compile_for_helper(const std::string& text, Variant variant, std::string& reply) {
    CodeObject code;
    std::string why;
    if (!compile(text, variant, code, why)) {
        reply = "ERR\n" + why;
        return false;
    }
    reply = "OK\n";
    for (int k = 0; k < kKernels; ++k) reply += code.kernel[k] + "\n";
    reply += std::to_string(code.image.size()) + "\n";
    reply.append(code.image.data(), code.image.size());
    return true;
}

namespace {
std::string helper_path() {
    Dl_info info{};
    if (!dladdr(reinterpret_cast<void*>(&compile_for_helper), &info) || !info.dli_fname) return std::string();
    std::string path = info.dli_fname;
    const size_t slash = path.rfind('/');
    path = slash == std::string::npos ? std::string("rtx_compile") : path.substr(0, slash + 1) + "rtx_compile";
    return access(path.c_str(), X_OK) == 0 ? path : std::string();
}

// Runs the helper on (text, variant); the code object lands in the cache.  False with the reason.
bool compile_in_helper(Job& job, std::string& why) {
    const std::string path = helper_path();
    if (path.empty()) {
        why = "the compile helper (rtx_compile beside librtx.so) is missing";
        return false;
    }
    int to_child[2], from_child[2];  // a socket pair towards the helper (a write to a helper that died is an error, not a signal)
    if (socketpair(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0, to_child) != 0) {
        why = "socketpair failed";
        return false;
    }
    if (pipe2(from_child, O_CLOEXEC) != 0) {
        close(to_child[0]);
        close(to_child[1]);
        why = "pipe2 failed";
        return false;
    }
    posix_spawn_file_actions_t actions;
    posix_spawn_file_actions_init(&actions);
    posix_spawn_file_actions_adddup2(&actions, to_child[0], 0);
    posix_spawn_file_actions_adddup2(&actions, from_child[1], 1);
    char* const argv[] = {const_cast<char*>(path.c_str()), nullptr};
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, path.c_str(), &actions, nullptr, argv, environ);
    posix_spawn_file_actions_destroy(&actions);
    close(to_child[0]);
    close(from_child[1]);
    if (rc != 0) {
        close(to_child[1]);
        close(from_child[0]);
        why = std::string("cannot start ") + path + ": " + std::strerror(rc);
        return false;
    }
    job.helper_pid.store((int)pid, std::memory_order_release);
    const std::string request = std::string(job.variant.light_tree ? "1 " : "0 ") + (job.variant.list_lit ? "1\n" : "0\n") + job.text;
    size_t sent = 0;
    while (sent < request.size()) {
        const ssize_t n = send(to_child[1], request.data() + sent, request.size() - sent, MSG_NOSIGNAL);
        if (n <= 0) break;
        sent += (size_t)n;
    }
    shutdown(to_child[1], SHUT_WR);
    std::string reply;
    char buf[65536];
    for (;;) {
        const ssize_t n = read(from_child[0], buf, sizeof buf);
        if (n <= 0) break;
        reply.append(buf, (size_t)n);
    }
    close(to_child[1]);
    close(from_child[0]);
    int status = 0;
    (void)waitpid(pid, &status, 0);
    job.helper_pid.store(0, std::memory_order_release);
    if (reply.compare(0, 3, "OK\n") != 0) {
        why = reply.compare(0, 4, "ERR\n") == 0 ? reply.substr(4, 600) : std::string("the compile helper ended without an answer");
        return false;
    }
    auto code = std::make_shared<CodeObject>();
    size_t at = 3;
    auto line = [&](std::string& out) {
        const size_t end = reply.find('\n', at);
        if (end == std::string::npos) return false;
        out = reply.substr(at, end - at);
        at = end + 1;
        return true;
    };
    std::string size_text;
    bool ok = true;
    for (int k = 0; k < kKernels; ++k) ok = ok && line(code->kernel[k]);
    ok = ok && line(size_text);
    const size_t size = ok ? (size_t)std::strtoull(size_text.c_str(), nullptr, 10) : 0;
    if (!ok || size == 0 || reply.size() - at != size) {
        why = "the compile helper's answer is malformed";
        return false;
    }
    code->image.assign(reply.begin() + (long)at, reply.end());
    insert_code(cache_key(job.text, job.variant), code);
    return true;
}
}  // namespace

std::shared_ptr<Job> start(const std::string& text, Variant variant) {
    auto job = std::make_shared<Job>();
    job->text = text;
    job->variant = variant;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        if (g_cache.count(cache_key(text, variant))) {  // (already compiled in this process: nothing to start)
            job->state.store(1, std::memory_order_release);
            return job;
        }
    }
    std::shared_ptr<Job> keep = job;  // (the thread holds the job: its owner may drop it at any time)
    job->thread = std::thread([keep] {
        const auto t0 = std::chrono::steady_clock::now();
        std::string why;
        const bool ok = compile_in_helper(*keep, why);
        keep->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        keep->why = why;
        keep->state.store(ok ? 1 : 2, std::memory_order_release);
    });
    return job;
}

void finish(const std::shared_ptr<Job>& job) {
    if (!job || !job->thread.joinable()) return;
    if (job->state.load(std::memory_order_acquire) == 0) {  // still compiling: its owner has no use for the result
        const int pid = job->helper_pid.load(std::memory_order_acquire);
        if (pid > 0) (void)kill((pid_t)pid, SIGTERM);
    }
    job->thread.join();
}

void unload(Kernel& k) {
    if (k.module) {
        (void)hipDeviceSynchronize();  // (a launch of these kernels may still be in flight on a caller's stream)
        (void)hipModuleUnload(k.module);
    }
    k = Kernel{};
}

}  // namespace rtxspecial
