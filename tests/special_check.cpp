// special_check.cpp — the failure path of the run-time compilation (host only, no GPU): a scan text that
// does not compile must come back as "false" with the compiler's message, not as an exception or a
// crash, because rtx_upload_scene then keeps the table-driven kernels and reports the reason.
// Built by tests/test_host_api.py with hipcc against raytrace_amd/csrc/rtx_special.cpp + rtx_sources.cpp.
#include <cstdio>
#include <string>

#include "rtx_special.h"

int main() {
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
    std::printf("OK reason: %.120s\n", why.c_str());
    return 0;
}
