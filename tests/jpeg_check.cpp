// jpeg_check.cpp - the JPEG decoder of the host library under the sanitizers: decodes the files
// named on the command line, then thousands of damaged copies of each (bytes overwritten, runs
// removed, the file cut short).  A damaged file may decode to anything or be refused; it must
// not read or write out of bounds, overflow a signed integer or allocate without bound.
//   usage: jpeg_check file.jpg ...   (prints "ok <decoded> <refused>")
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "image_decode.h"

static uint32_t g_state = 12345u;
static uint32_t rnd() {
    g_state ^= g_state << 13;
    g_state ^= g_state >> 17;
    g_state ^= g_state << 5;
    return g_state;
}

// Decodes a copy of `bytes` held in an allocation of exactly bytes.size() bytes: a std::vector that was
// shrunk keeps its capacity, and the sanitizer does not see reads between size() and capacity().
static int try_file(const std::vector<unsigned char>& bytes, long& decoded, long& refused) {
    const std::vector<unsigned char> exact(bytes.begin(), bytes.end());
    int bw = 0, bh = 0;
    std::vector<unsigned char> out;
    std::string err;
    if (rt::detail::load_jpeg(exact, bw, bh, out, err)) {
        if (out.size() != (size_t)bw * bh * 3) return 2;
        ++decoded;
    } else {
        if (err.empty()) return 3;
        ++refused;
    }
    return 0;
}

static size_t find_marker(const std::vector<unsigned char>& f, unsigned char m) {
    for (size_t i = 2; i + 1 < f.size(); ++i)
        if (f[i] == 0xFF && f[i + 1] == m) return i;
    return 2;
}

int main(int argc, char** argv) {
    long decoded = 0, refused = 0;
    for (int a = 1; a < argc; ++a) {
        std::vector<unsigned char> file;
        FILE* fp = std::fopen(argv[a], "rb");
        if (!fp) {
            std::fprintf(stderr, "cannot open %s\n", argv[a]);
            return 1;
        }
        unsigned char buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, fp)) > 0) file.insert(file.end(), buf, buf + n);
        std::fclose(fp);
        int w = 0, h = 0;
        std::vector<unsigned char> rgb;
        std::string why;
        if (!rt::detail::load_jpeg(file, w, h, rgb, why) || rgb.size() != (size_t)w * h * 3) {
            std::fprintf(stderr, "%s: intact file refused: %s\n", argv[a], why.c_str());
            return 1;
        }
        for (int trial = 0; trial < 3000; ++trial) {
            std::vector<unsigned char> bad = file;
            const uint32_t kind = rnd() % 4u;
            if (kind == 0u) {  // a few bytes overwritten
                for (uint32_t k = 0, m = 1u + rnd() % 4u; k < m; ++k) bad[rnd() % bad.size()] = (unsigned char)rnd();
            } else if (kind == 1u) {  // cut short
                bad.resize(rnd() % bad.size());
            } else if (kind == 2u) {  // a run removed
                const size_t at = rnd() % bad.size(), len = 1u + rnd() % 16u;
                bad.erase(bad.begin() + at, bad.begin() + (at + len < bad.size() ? at + len : bad.size()));
            } else {  // damage confined to the headers, where the sizes and table indices live
                const size_t head = bad.size() < 700 ? bad.size() : 700;
                bad[rnd() % head] = (unsigned char)rnd();
            }
            const int rc = try_file(bad, decoded, refused);
            if (rc) return rc;
        }
        // the marker walk at the end of a file: runs of fill bytes (0xFF) up to the last byte, and a
        // marker behind fill bytes with fewer than the two length bytes left
        const size_t sos = find_marker(file, 0xDA);
        for (size_t keep : {(size_t)2, sos, file.size() - 2}) {
            for (int fill = 1; fill <= 6; ++fill) {
                std::vector<unsigned char> bad(file.begin(), file.begin() + keep);
                bad.insert(bad.end(), (size_t)fill, 0xFF);
                int rc = try_file(bad, decoded, refused);
                if (rc) return rc;
                for (unsigned char marker : {0xDB, 0xC4, 0xC0, 0xDA, 0xD9, 0xE0, 0x00}) {
                    for (int tail = 0; tail <= 2; ++tail) {
                        std::vector<unsigned char> b2 = bad;
                        b2.push_back(marker);
                        b2.insert(b2.end(), (size_t)tail, 0x00);
                        rc = try_file(b2, decoded, refused);
                        if (rc) return rc;
                    }
                }
            }
        }
    }
    std::printf("ok %ld %ld\n", decoded, refused);
    return 0;
}
