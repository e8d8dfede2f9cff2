// rtx_compile — the compile helper of librtx.so: reads "<light_tree> <list_lit>\\n" and the text of a
// scene's kernels from standard input, compiles them with the HIP run-time compiler and writes the
// code object to standard output (format: rtx_special.cpp).  Started by the library for compilations in
// the background; touches no GPU.
#include <cstdio>
#include <string>

#include <unistd.h>

#include "rtx_special.h"

int main() {
    std::string request;
    char buf[65536];
    for (;;) {
        const ssize_t n = read(0, buf, sizeof buf);
        if (n <= 0) break;
        request.append(buf, (size_t)n);
    }
    const size_t eol = request.find('\n');
    std::string reply;
    bool ok = false;
    if (eol == std::string::npos || eol < 3) {
        reply = "ERR\nmalformed request";
    } else {
        rtxspecial::Variant variant;
        variant.light_tree = request[0] == '1';
        variant.list_lit = request[2] == '1';
        ok = rtxspecial::compile_for_helper(request.substr(eol + 1), variant, reply);
    }
    size_t sent = 0;
    while (sent < reply.size()) {
        const ssize_t n = write(1, reply.data() + sent, reply.size() - sent);
        if (n <= 0) return 2;
        sent += (size_t)n;
    }
    return ok ? 0 : 1;
}
