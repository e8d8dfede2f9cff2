// main.cpp — command-line entry: the counterpart of the reference's main()
// (Raytrace_C++/Src/main.cpp:7-20) without its Windows shell glue.  Renders a scene
// on the GPU and writes Output/<name>.bmp.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" int rtxh_render_to_bmp(const char* name, const char* path, int w, int h, int spp, int deterministic,
                                  unsigned long long seed, int device, unsigned long long* rays, double* seconds);
extern "C" int rtxh_set_kernel_compilation(int mode);
extern "C" int rtxh_render_to_bmp_multi(const char* name, const char* path, int w, int h, int spp, int deterministic,
                                        unsigned long long seed, const int* devices, int n_devices,
                                        unsigned long long* rays, double* seconds);
extern "C" const char* rtxh_last_error(void);

int main(int argc, char** argv) {
    // reference defaults: 800 x 400, 50 spp, Output/40_Test.bmp (main.cpp:8-11)
    int nx = 800, ny = 400, ns = 50, det = 0, device = 0;
    unsigned long long seed = 0;
    std::string scene = "cornell_builtin", out = "Output/40_Test.bmp";
    std::vector<int> devices;  // --devices 0,1,2,...: the frame split by sample range over these GPUs
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", what); std::exit(2); }
            return argv[++i];
        };
        if (a == "--scene") scene = next("--scene");
        else if (a == "--out") out = next("--out");
        else if (a == "--width") nx = std::atoi(next("--width"));
        else if (a == "--height") ny = std::atoi(next("--height"));
        else if (a == "--spp") ns = std::atoi(next("--spp"));
        else if (a == "--seed") seed = std::strtoull(next("--seed"), nullptr, 10);
        else if (a == "--device") device = std::atoi(next("--device"));
        else if (a == "--devices") {
            const char* list = next("--devices");
            for (const char* q = list; *q;) {
                char* end = nullptr;
                devices.push_back((int)std::strtol(q, &end, 10));
                if (end == q) { std::fprintf(stderr, "bad --devices list '%s'\n", list); return 2; }
                q = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "--deterministic") det = 1;
        else if (a == "--compile") {
            const std::string m = next("--compile");
            const int mode = m == "auto" ? -1 : m == "never" ? 0 : m == "upload" ? 1 : m == "background" ? 2 : 3;
            if (!rtxh_set_kernel_compilation(mode)) { std::fprintf(stderr, "bad --compile '%s' (auto, never, upload, background)\n", m.c_str()); return 2; }
        }
        else {
            std::fprintf(stderr,
                         "usage: %s [--scene NAME] [--out FILE.bmp] [--width W] [--height H] [--spp N]\n"
                         "          [--seed S] [--device D | --devices D0,D1,...] [--deterministic]\n"
                         "          [--compile auto|never|upload|background]   (kernels compiled for the scene)\n", argv[0]);
            return 2;
        }
    }
    unsigned long long rays = 0;
    double seconds = 0.0;
    const int ok = devices.empty()
                       ? rtxh_render_to_bmp(scene.c_str(), out.c_str(), nx, ny, ns, det, seed, device, &rays, &seconds)
                       : rtxh_render_to_bmp_multi(scene.c_str(), out.c_str(), nx, ny, ns, det, seed, devices.data(),
                                                  (int)devices.size(), &rays, &seconds);
    if (!ok) {
        std::fprintf(stderr, "render failed: %s\n", rtxh_last_error());
        return 1;
    }
    std::printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"spp\": %d, \"rays\": %llu, "
                "\"seconds\": %.6f, \"mrays_per_s\": %.3f, \"file\": \"%s\"}\n",
                scene.c_str(), nx, ny, ns, rays, seconds, seconds > 0 ? rays / seconds * 1e-6 : 0.0, out.c_str());
    return 0;
}
