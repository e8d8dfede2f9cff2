// special_check.cpp — the run-time compilation on the host (no GPU).  The failure path: a scan text that
// does not compile must come back as "false" with the compiler's message, not as an exception or a
// crash, because rtx_upload_scene then keeps the table-driven kernels and reports the reason.  The
// generators: feature set, scan and light list of a hand-made table, whose text must compile.
// Built by tests/test_host_api.py with hipcc against raytrace_amd/csrc/rtx_special.cpp + rtx_sources.cpp.
#include <chrono>
#include <cstdio>
#include <thread>
#include <string>
#include <vector>

#include "rtx_special.h"

int main(int argc, char** argv) {
    std::string why;
    double seconds = -1.0;
    const std::string bad = "#define RTX_SPECIAL_SCAN this is not C++;\n";
    if (rtxspecial::prepare(bad, rtxspecial::Variant{}, why, &seconds)) {
        std::printf("FAIL: a scan that cannot compile was accepted\n");
        return 1;
    }
    if (why.find("run-time compilation failed") == std::string::npos && why.find("not available") == std::string::npos) {
        std::printf("FAIL: no reason given: '%s'\n", why.c_str());
        return 1;
    }
    // the failure is not cached: the same text fails again, and a good text still compiles afterwards
    std::string again;
    if (rtxspecial::prepare(bad, rtxspecial::Variant{}, again, nullptr)) {
        std::printf("FAIL: accepted on the second call\n");
        return 1;
    }
    std::string ok_why;
    const std::string good = "#define RTX_SPECIAL_SCAN { RTX_SPECIAL_GROUP_IDENTITY(); RTX_SPECIAL_SPHERES(); "
                             "RTX_SPECIAL_SPHERE(0x00000000u, 0x00000000u, 0x00000000u, 0x3f800000u, 0u); }\n";
    if (why.find("not available") == std::string::npos && !rtxspecial::prepare(good, rtxspecial::Variant{}, ok_why, &seconds)) {
        std::printf("FAIL: a one-sphere scan did not compile: %s\n", ok_why.c_str());
        return 1;
    }
    // the generators on a hand-made table: an identity group with one rect of each kind and a sphere, a
    // rotated group with a box (six slots), two lights; what they write must compile
    std::vector<rtxd::ScanGroup> groups(2);
    std::vector<rtxd::ScanRec> recs;
    std::vector<rtxd::ScanHit> hits;
    auto rec = [&](float a, float b, float c, float d, float e, float f, uint32_t mat) {
        rtxd::ScanRec R{};
        R.a = a; R.b = b; R.c = c; R.d = d; R.e = e; R.f = f;
        recs.push_back(R);
        rtxd::ScanHit H{};
        H.bits = mat << rtxd::kInfoMatShift;
        H.param = mat == RTX_MAT_METAL ? 0.25f : 1.5f;
        hits.push_back(H);
    };
    groups[0].identity = 1;
    groups[0].first = 0;
    groups[0].n_xy = groups[0].n_xz = groups[0].n_yz = groups[0].n_sph = 1;
    rec(0.0f, 5.0f, 0.0f, 5.0f, 5.0f, 0.0f, RTX_MAT_LAMBERTIAN);
    rec(-1.0f, 2.0f, 1.0f, 3.0f, 4.5f, 0.0f, RTX_MAT_DIFFUSE_LIGHT);
    rec(0.0f, 5.0f, -2.0f, 5.0f, 0.0f, 0.0f, RTX_MAT_METAL);
    rec(1.0f, 1.0f, 1.0f, 0.5f, 0.25f, 0.0f, RTX_MAT_DIELECTRIC);
    groups[1].rot_y = 1;
    groups[1].first = (uint32_t)recs.size();
    groups[1].n_box = 1;
    const float m[12] = {0.8f, 0.0f, -0.6f, 0.0f, 1.0f, 0.0f, 0.6f, 0.0f, 0.8f, -1.0f, 0.0f, -2.0f};
    for (int k = 0; k < 12; ++k) groups[1].m[k] = m[k];
    for (uint32_t f = 0; f < rtxd::kBoxSlots; ++f) rec(0.0f, 0.0f, 0.0f, 1.0f, 2.0f, 1.0f, RTX_MAT_LAMBERTIAN);
    recs.push_back(recs.back());  // the table's pad record
    std::vector<rtxd::Light> lights(2);
    lights[0].a = -1.0f; lights[0].b = 2.0f; lights[0].c = 1.0f; lights[0].d = 3.0f; lights[0].e = 4.5f;
    lights[0].kind = RTX_LIGHT_RECT_XZ; lights[0].inv_area = 1.0f / 6.0f; lights[0].pad = 0;
    lights[1] = lights[0];
    lights[1].kind = RTX_LIGHT_SPHERE;
    rtxspecial::Variant list_lit;
    list_lit.list_lit = true;
    const std::string text = rtxspecial::features_source(hits, lights, 2, false) + rtxspecial::scan_source(groups, recs) +
                             rtxspecial::lights_source(lights, 2, true);
    if (text.find("RTX_SPECIAL_FEATURES 0x1fc") == std::string::npos || text.find("RTX_SPECIAL_RANGE0, RTX_SPECIAL_RANGE0") == std::string::npos ||
        text.find("RTX_SPECIAL_BOX(") == std::string::npos || text.find("RTX_SPECIAL_LIGHT_WEIGHT 0x3f000000u") == std::string::npos) {
        std::printf("FAIL: generated text\n%s\n", text.c_str());
        return 1;
    }
    std::string gen_why;
    if (why.find("not available") == std::string::npos && !rtxspecial::prepare(text, list_lit, gen_why, nullptr)) {
        std::printf("FAIL: the generated text did not compile: %s\n", gen_why.c_str());
        return 1;
    }
    // compilations in the background (the helper rtx_compile must lie beside this program): one that succeeds,
    // one that fails with the compiler's message, one whose owner gives up at once
    if (argc > 1 && std::string(argv[1]) == "--jobs" && why.find("not available") == std::string::npos) {
        auto wait_for = [](const std::shared_ptr<rtxspecial::Job>& job) {
            for (int k = 0; k < 3000 && job->state.load() == 0; ++k) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            return job->state.load();
        };
        const std::string fresh = good + "// a text this process has not compiled yet\n";
        auto ok_job = rtxspecial::start(fresh, rtxspecial::Variant{});
        auto bad_job = rtxspecial::start(bad + "// again\n", rtxspecial::Variant{});
        if (wait_for(ok_job) != 1) {
            std::printf("FAIL: background job: state %d, %s\n", ok_job->state.load(), ok_job->why.c_str());
            return 1;
        }
        if (wait_for(bad_job) != 2 || bad_job->why.find("run-time compilation failed") == std::string::npos) {
            std::printf("FAIL: a failing background job reports state %d, '%s'\n", bad_job->state.load(), bad_job->why.c_str());
            return 1;
        }
        rtxspecial::finish(ok_job);
        rtxspecial::finish(bad_job);
        std::string cached_why;
        double seconds = -1.0;
        if (!rtxspecial::prepare(fresh, rtxspecial::Variant{}, cached_why, &seconds) || seconds != 0.0) {
            std::printf("FAIL: the helper's code object is not in the cache\n");
            return 1;
        }
        auto again = rtxspecial::start(fresh, rtxspecial::Variant{});  // already in the cache: done at once, no thread
        if (again->state.load() != 1 || again->thread.joinable()) {
            std::printf("FAIL: a cached text started a compilation\n");
            return 1;
        }
        const auto t0 = std::chrono::steady_clock::now();
        auto dropped = rtxspecial::start(good + "// given up\n", rtxspecial::Variant{});
        std::this_thread::sleep_for(std::chrono::milliseconds(30));
        rtxspecial::finish(dropped);  // ends the helper instead of waiting for it
        const double gave_up = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (gave_up > 5.0) {
            std::printf("FAIL: finish() waited %.1f s\n", gave_up);
            return 1;
        }
        auto left = rtxspecial::start(good + "// never finished by its owner\n", rtxspecial::Variant{});
        (void)left;  // the process ends with this one still compiling
        std::printf("OK jobs\n");
        return 0;
    }
    std::printf("OK reason: %.120s\n", why.c_str());
    return 0;
}
