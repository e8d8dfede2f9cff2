// jpeg_baseline.cpp — decoder for sequential Huffman JPEG files (SOF0 / SOF1, 8 bits), the
// format of the reference's one texture asset (Asset/Texture/field001.jpg, 160 x 160, 4:2:0).
//
// The reference reads its textures with stb_image v2.30 (ImageTexture.h:9, stbi_load with
// req_comp = 0), a third-party header it vendors under external/stb/.  A JPEG decoder is free
// to round its inverse DCT, chroma upsampling and colour conversion as it likes, so "the same
// texels as the reference" means the texels THAT decoder produces.  This file therefore
// follows stb_image's arithmetic step by step (cited below by stb_image.h line) in its own
// structure: planes in std::vector, one bit reader, a table-free canonical Huffman walk, and a
// row walker for the upsampling instead of function pointers.  tests/test_oracle_golden.py
// compares the texels with the ones the reference's own stbi_load returns.
//
// Not handled (load fails, the caller reports it): progressive and arithmetic-coded files,
// 12-bit samples, CMYK / YCCK.  Grey-scale files come back as R = G = B.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "image_decode.h"

namespace rt {
namespace detail {
namespace {

// zig-zag position -> row-major coefficient index
const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Canonical Huffman code of one DHT segment: codes of length L are consecutive, starting at
// first_code[L]; their symbols are symbols[first_index[L] ...].
struct Huffman {
    bool defined = false;
    int count[17] = {0};
    int first_code[17] = {0};
    int first_index[17] = {0};
    uint8_t symbols[256] = {0};

    bool build(const uint8_t* counts16, const uint8_t* syms, int n_syms) {
        int code = 0, index = 0;
        for (int len = 1; len <= 16; ++len) {
            count[len] = counts16[len - 1];
            first_code[len] = code;
            first_index[len] = index;
            code += count[len];
            index += count[len];
            if (code > (1 << len)) return false;  // more codes of this length than fit
            code <<= 1;
        }
        if (index != n_syms || index > 256) return false;
        std::memcpy(symbols, syms, (size_t)n_syms);
        defined = true;
        return true;
    }
};

// Entropy-coded segment reader: bytes of the scan with the stuffed zero after 0xFF removed; at
// a marker the reader stops consuming and supplies zero bits (stb_image.h:2075-2095).
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int bits = 0;
    int marker = -1;  // marker met inside the scan (0xFF xx), -1 while none

    BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}

    void fill() {
        while (bits <= 24) {
            uint32_t byte = 0;
            if (marker < 0 && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    int next = p < end ? *p : 0xD9;
                    while (next == 0xFF && p + 1 < end) next = *++p;  // fill bytes
                    if (next == 0) {
                        ++p;  // stuffed zero: a literal 0xFF
                    } else {
                        marker = next;
                        if (p < end) ++p;
                        byte = 0;
                    }
                }
            }
            acc |= byte << (24 - bits);
            bits += 8;
        }
    }
    int bit() {
        if (bits < 1) fill();
        const int b = (int)(acc >> 31);
        acc <<= 1;
        --bits;
        return b;
    }
    // n raw bits, n in 0 .. 16
    int take(int n) {
        if (n == 0) return 0;
        if (bits < n) fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n;
        bits -= n;
        return v;
    }
    void restart() {
        acc = 0;
        bits = 0;
        marker = -1;
    }
};

// One Huffman symbol, or -1 for a code the table does not hold.
int decode_symbol(BitReader& br, const Huffman& h) {
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
        code = (code << 1) | br.bit();
        const int offset = code - h.first_code[len];
        if (offset >= 0 && offset < h.count[len]) return h.symbols[h.first_index[len] + offset];
    }
    return -1;
}

// The signed value of an n-bit magnitude category (ITU T.81 F.2.2.1 EXTEND).
int receive_extend(BitReader& br, int n) {
    if (n == 0) return 0;
    const int v = br.take(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;   // Huffman table selectors of the current scan
    int width = 0, height = 0;  // samples that belong to the image
    int stride = 0, rows = 0;   // padded to whole MCUs
    int pred = 0;               // DC predictor
    std::vector<uint8_t> plane;
};

// One 8x8 block: DC difference and run-length coded AC terms, each multiplied by its
// quantiser and kept as 16-bit values as the reference's decoder keeps them
// (stb_image.h:2210-2265).
bool decode_block(BitReader& br, const Huffman& dc, const Huffman& ac, const uint16_t* quant, int& pred,
                  int16_t coef[64]) {
    std::memset(coef, 0, 64 * sizeof(int16_t));
    const int t = decode_symbol(br, dc);
    if (t < 0 || t > 15) return false;
    pred = (int)((uint32_t)pred + (uint32_t)receive_extend(br, t));  // (wraps on a corrupt file)
    coef[0] = (int16_t)((uint32_t)pred * quant[0]);
    for (int k = 1; k < 64;) {
        const int rs = decode_symbol(br, ac);
        if (rs < 0) return false;
        const int run = rs >> 4, size = rs & 15;
        if (size == 0) {
            if (rs != 0xF0) break;  // end of block
            k += 16;
            continue;
        }
        k += run;
        if (k > 63) return false;
        const int pos = kZigZag[k++];
        coef[pos] = (int16_t)(receive_extend(br, size) * quant[pos]);
    }
    return true;
}

// The 8-point inverse DCT both passes share (the "islow" factorisation of libjpeg's
// jidctint.c, constants in 12-bit fixed point rounded as stb_image.h:2426-2465 rounds them).
// out: the even part sums e[0..3] and the odd part o[0..3], the results being e[i] + o[3-i]
// (rows / columns 0 .. 3) and e[i] - o[3-i] (7 .. 4).  The arithmetic is 32-bit two's complement
// carried out on unsigned words: the same bits as the reference's int arithmetic wherever that
// does not overflow, and defined (wrapping) where a corrupt file makes it overflow.
using word = uint32_t;
inline word fix(double x) { return (word)(int)(x * 4096 + 0.5); }
inline int sra(word x, int n) { return (int)(int32_t)x >> n; }  // arithmetic shift of the signed value
struct Idct1D {
    word e[4], o[4];
    void run(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7) {
        const word s0 = (word)i0, s1 = (word)i1, s2 = (word)i2, s3 = (word)i3, s4 = (word)i4, s5 = (word)i5, s6 = (word)i6,
                   s7 = (word)i7;
        // even part: rotation of (s2, s6), butterflies with (s0 +- s4) << 12
        const word z = (s2 + s6) * fix(0.5411961f);
        const word r6 = z + s6 * fix(-1.847759065f);
        const word r2 = z + s2 * fix(0.765366865f);
        const word a = (s0 + s4) * 4096u, b = (s0 - s4) * 4096u;
        e[0] = a + r2;
        e[3] = a - r2;
        e[1] = b + r6;
        e[2] = b - r6;
        // odd part
        const word q3 = s7 + s3, q4 = s5 + s1, q1 = s7 + s1, q2 = s5 + s3;
        const word q5 = (q3 + q4) * fix(1.175875602f);
        const word m1 = q5 + q1 * fix(-0.899976223f), m2 = q5 + q2 * fix(-2.562915447f);
        const word m3 = q3 * fix(-1.961570560f), m4 = q4 * fix(-0.390180644f);
        o[3] = s1 * fix(1.501321110f) + m1 + m4;
        o[2] = s3 * fix(3.072711026f) + m2 + m3;
        o[1] = s5 * fix(2.053119869f) + m2 + m4;
        o[0] = s7 * fix(0.298631336f) + m1 + m3;
    }
};

uint8_t clamp_u8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// Columns first (two fractional bits kept, a column of zero AC terms short-cut to 4 * DC),
// then rows with the level shift and the rounding folded into one constant
// (stb_image.h:2467-2524).
void idct_block(const int16_t coef[64], uint8_t* out, int stride) {
    int tmp[64];
    Idct1D t;
    for (int c = 0; c < 8; ++c) {
        const int16_t* d = coef + c;
        if (!(d[8] | d[16] | d[24] | d[32] | d[40] | d[48] | d[56])) {
            const int dc = d[0] * 4;
            for (int r = 0; r < 8; ++r) tmp[8 * r + c] = dc;
            continue;
        }
        t.run(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56]);
        for (int i = 0; i < 4; ++i) {
            const word ev = t.e[i] + 512u;
            tmp[8 * i + c] = sra(ev + t.o[3 - i], 10);
            tmp[8 * (7 - i) + c] = sra(ev - t.o[3 - i], 10);
        }
    }
    for (int r = 0; r < 8; ++r) {
        const int* v = tmp + 8 * r;
        t.run(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        uint8_t* o = out + (size_t)r * stride;
        for (int i = 0; i < 4; ++i) {
            const word ev = t.e[i] + 65536u + (128u << 17);
            o[i] = clamp_u8(sra(ev + t.o[3 - i], 17));
            o[7 - i] = clamp_u8(sra(ev - t.o[3 - i], 17));
        }
    }
}

// One output row of a component whose plane is hs x vs times smaller than the image
// (stb_image.h:3465-3527, 3646-3655): triangle filter for factors of two - weights 3:1
// between the two nearest samples along each axis -, sample replication for any other factor.
void upsample_row(const uint8_t* near_row, const uint8_t* far_row, int w, int hs, int vs, uint8_t* out) {
    if (hs == 1 && vs == 1) {
        std::memcpy(out, near_row, (size_t)w);
    } else if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near_row[i] + far_row[i] + 2) >> 2);
    } else if (hs == 2 && vs == 1) {
        if (w == 1) {
            out[0] = out[1] = near_row[0];
            return;
        }
        out[0] = near_row[0];
        out[2 * w - 1] = near_row[w - 1];
        for (int i = 0; i + 1 < w; ++i) {  // the two outputs between samples i and i + 1
            out[2 * i + 1] = (uint8_t)((3 * near_row[i] + near_row[i + 1] + 2) >> 2);
            out[2 * i + 2] = (uint8_t)((3 * near_row[i + 1] + near_row[i] + 2) >> 2);
        }
        // the reference's decoder weights the last pair the other way round (stb_image.h:3494)
        out[2 * w - 2] = (uint8_t)((3 * near_row[w - 2] + near_row[w - 1] + 2) >> 2);
    } else if (hs == 2 && vs == 2) {
        // vertical blend first (kept at four times the scale), then the horizontal one
        int cur = 3 * near_row[0] + far_row[0];
        if (w == 1) {
            out[0] = out[1] = (uint8_t)((cur + 2) >> 2);
            return;
        }
        out[0] = (uint8_t)((cur + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int prev = cur;
            cur = 3 * near_row[i] + far_row[i];
            out[2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);
            out[2 * i] = (uint8_t)((3 * cur + prev + 8) >> 4);
        }
        out[2 * w - 1] = (uint8_t)((cur + 2) >> 2);
    } else {
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < hs; ++j) out[i * hs + j] = near_row[i];
    }
}

// Walks the rows of one component in step with the image rows: which plane row is the nearer
// one and which the farther one for every output row (stb_image.h:3907-3940: the phase starts
// in the middle of the first sample and the last sample row repeats below the image).
struct RowWalker {
    int vs, phase, row = 0, rows;
    int line0 = 0, line1 = 0;
    RowWalker(int vs_, int rows_) : vs(vs_), phase(vs_ >> 1), rows(rows_) {}
    void current(int& near_row, int& far_row) const {
        const bool lower = phase >= (vs >> 1);
        near_row = lower ? line1 : line0;
        far_row = lower ? line0 : line1;
    }
    void advance() {
        if (++phase >= vs) {
            phase = 0;
            line0 = line1;
            if (++row < rows) ++line1;
        }
    }
};

// Fixed-point YCbCr -> RGB with the constants and the masking of the reference's decoder
// (stb_image.h:3659-3683).
inline int fix20(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }
void ycc_to_rgb(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int n, uint8_t* out) {
    for (int i = 0; i < n; ++i) {
        const int yf = (y[i] << 20) + (1 << 19);
        const int r_ = cr[i] - 128, b_ = cb[i] - 128;
        const int r = (yf + r_ * fix20(1.40200f)) >> 20;
        const int g = (yf + r_ * -fix20(0.71414f) + (int)((uint32_t)(b_ * -fix20(0.34414f)) & 0xffff0000u)) >> 20;
        const int b = (yf + b_ * fix20(1.77200f)) >> 20;
        out[3 * i + 0] = clamp_u8(r);
        out[3 * i + 1] = clamp_u8(g);
        out[3 * i + 2] = clamp_u8(b);
    }
}

struct Decoder {
    const std::vector<unsigned char>& d;
    std::string& why;
    uint16_t quant[4][64] = {};
    bool quant_defined[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    std::vector<Component> comps;
    int width = 0, height = 0, h_max = 1, v_max = 1, mcus_x = 0, mcus_y = 0;
    int restart_interval = 0;
    bool jfif = false;
    int adobe_transform = -1;
    bool frame_seen = false;

    Decoder(const std::vector<unsigned char>& data, std::string& w) : d(data), why(w) {}
    bool fail(const char* m) {
        why = m;
        return false;
    }

    bool read_dqt(size_t p, size_t end) {
        while (p < end) {
            const int pq = d[p] >> 4, tq = d[p] & 15;
            ++p;
            if (pq > 1 || tq > 3) return fail("bad quantisation table header");
            const size_t need = pq ? 128 : 64;
            if (p + need > end) return fail("quantisation table cut short");
            for (int i = 0; i < 64; ++i) {
                quant[tq][kZigZag[i]] = pq ? (uint16_t)((d[p] << 8) | d[p + 1]) : d[p];
                p += pq ? 2 : 1;
            }
            quant_defined[tq] = true;
        }
        return true;
    }

    bool read_dht(size_t p, size_t end) {
        while (p < end) {
            if (p + 17 > end) return fail("Huffman table cut short");
            const int tc = d[p] >> 4, th = d[p] & 15;
            if (tc > 1 || th > 3) return fail("bad Huffman table header");
            int n = 0;
            for (int i = 0; i < 16; ++i) n += d[p + 1 + i];
            if (n > 256 || p + 17 + (size_t)n > end) return fail("Huffman table cut short");
            if (!(tc ? ac[th] : dc[th]).build(&d[p + 1], &d[p + 17], n)) return fail("bad Huffman code lengths");
            p += 17 + (size_t)n;
        }
        return true;
    }

    bool read_sof(size_t p, size_t end) {
        if (frame_seen) return fail("more than one frame");
        if (end - p < 6) return fail("frame header cut short");
        if (d[p] != 8) return fail("only 8-bit samples are supported");
        height = (d[p + 1] << 8) | d[p + 2];
        width = (d[p + 3] << 8) | d[p + 4];
        const int n = d[p + 5];
        if (width <= 0 || height <= 0) return fail("empty frame");
        if ((uint64_t)width * (uint64_t)height > (1ull << 26)) return fail("frame larger than 2^26 pixels");
        if (n != 1 && n != 3) return fail("only grey-scale and three-component files are supported");
        if (end - p < 6 + 3 * (size_t)n) return fail("frame header cut short");
        comps.resize((size_t)n);
        for (int k = 0; k < n; ++k) {
            Component& c = comps[(size_t)k];
            c.id = d[p + 6 + 3 * k];
            c.h = d[p + 7 + 3 * k] >> 4;
            c.v = d[p + 7 + 3 * k] & 15;
            c.tq = d[p + 8 + 3 * k];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component header");
            h_max = c.h > h_max ? c.h : h_max;
            v_max = c.v > v_max ? c.v : v_max;
        }
        for (Component& c : comps)
            if (h_max % c.h || v_max % c.v) return fail("fractional sampling ratios are not supported");
        mcus_x = (width + 8 * h_max - 1) / (8 * h_max);
        mcus_y = (height + 8 * v_max - 1) / (8 * v_max);
        for (Component& c : comps) {
            c.width = (width * c.h + h_max - 1) / h_max;
            c.height = (height * c.v + v_max - 1) / v_max;
            c.stride = mcus_x * c.h * 8;
            c.rows = mcus_y * c.v * 8;
            c.plane.assign((size_t)c.stride * c.rows, 0);
        }
        frame_seen = true;
        return true;
    }

    // One scan: header at p (after the length), entropy-coded data from `data` on.  Returns the
    // position behind the scan's data in `next`.
    bool read_scan(size_t p, size_t end, size_t data, size_t& next) {
        if (!frame_seen) return fail("scan before frame header");
        if (end - p < 1) return fail("scan header cut short");
        const int ns = d[p];
        if (ns < 1 || ns > (int)comps.size() || end - p < 1 + 2 * (size_t)ns + 3) return fail("bad scan header");
        std::vector<Component*> in_scan;
        for (int k = 0; k < ns; ++k) {
            const int id = d[p + 1 + 2 * k], sel = d[p + 2 + 2 * k];
            Component* c = nullptr;
            for (Component& cand : comps)
                if (cand.id == id) c = &cand;
            if (!c) return fail("scan names an unknown component");
            c->td = sel >> 4;
            c->ta = sel & 15;
            if (c->td > 3 || c->ta > 3 || !dc[c->td].defined || !ac[c->ta].defined) return fail("scan uses an undefined Huffman table");
            if (!quant_defined[c->tq]) return fail("component uses an undefined quantisation table");
            c->pred = 0;
            in_scan.push_back(c);
        }
        const size_t q = p + 1 + 2 * (size_t)ns;
        if (d[q] != 0 || d[q + 1] != 63 || d[q + 2] != 0) return fail("not a sequential scan");

        BitReader br(d.data() + data, d.data() + d.size());
        int16_t coef[64];
        int until_restart = restart_interval ? restart_interval : 0x7fffffff;
        auto one_block = [&](Component& c, int bx, int by) {
            if (!decode_block(br, dc[c.td], ac[c.ta], quant[c.tq], c.pred, coef)) return false;
            idct_block(coef, &c.plane[(size_t)by * 8 * c.stride + (size_t)bx * 8], c.stride);
            return true;
        };
        auto after_unit = [&]() {  // restart markers sit between MCUs, byte aligned
            if (--until_restart > 0) return true;
            br.fill();
            if (br.marker < 0xD0 || br.marker > 0xD7) return false;  // the scan ends here
            br.restart();
            for (Component* c : in_scan) c->pred = 0;
            until_restart = restart_interval;
            return true;
        };
        bool more = true;
        if (ns == 1) {  // a scan of one component covers just that component's own blocks
            Component& c = *in_scan[0];
            const int bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
            for (int by = 0; by < bh && more; ++by)
                for (int bx = 0; bx < bw && more; ++bx) {
                    if (!one_block(c, bx, by)) return fail("bad entropy-coded data");
                    more = after_unit();
                }
        } else {
            for (int my = 0; my < mcus_y && more; ++my)
                for (int mx = 0; mx < mcus_x && more; ++mx) {
                    for (Component* c : in_scan)
                        for (int y = 0; y < c->v; ++y)
                            for (int x = 0; x < c->h; ++x)
                                if (!one_block(*c, mx * c->h + x, my * c->v + y)) return fail("bad entropy-coded data");
                    more = after_unit();
                }
        }
        // the next marker: either the one the bit reader ran into, or the first 0xFF xx (xx not 0,
        // not a restart marker) behind the bytes it consumed
        size_t pos = (size_t)(br.p - d.data());
        if (br.marker >= 0 && !(br.marker >= 0xD0 && br.marker <= 0xD7)) {
            next = pos - 2;
            return true;
        }
        while (pos + 1 < d.size()) {
            if (d[pos] == 0xFF && d[pos + 1] != 0 && d[pos + 1] != 0xFF && !(d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) break;
            ++pos;
        }
        next = pos;
        return true;
    }

    bool parse() {
        if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail("not a JPEG file");
        size_t p = 2;
        bool any_scan = false;
        while (p + 4 <= d.size()) {
            if (d[p] != 0xFF) return fail("marker expected");
            while (p + 1 < d.size() && d[p + 1] == 0xFF) ++p;  // fill bytes
            if (p + 2 > d.size()) break;  // the file ends in fill bytes
            const int m = d[p + 1];
            if (m == 0xD9) break;
            if (p + 4 > d.size()) return fail("segment cut short");  // the fill bytes used up the room for a length
            const size_t len = ((size_t)d[p + 2] << 8) | d[p + 3];
            const size_t body = p + 4, end = p + 2 + len;
            if (len < 2 || end > d.size()) return fail("segment cut short");
            size_t next = end;
            switch (m) {
                case 0xDB: if (!read_dqt(body, end)) return false; break;
                case 0xC4: if (!read_dht(body, end)) return false; break;
                case 0xC0: case 0xC1: if (!read_sof(body, end)) return false; break;
                case 0xC2: return fail("progressive JPEG files are not supported");
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    return fail("this JPEG coding process is not supported");
                case 0xDD:
                    if (len != 4) return fail("bad restart interval segment");
                    restart_interval = (d[body] << 8) | d[body + 1];
                    break;
                case 0xE0:
                    if (len >= 7 && !std::memcmp(&d[body], "JFIF", 5)) jfif = true;
                    break;
                case 0xEE:
                    if (len >= 14 && !std::memcmp(&d[body], "Adobe", 6)) adobe_transform = d[body + 11];
                    break;
                case 0xDA:
                    if (!read_scan(body, end, end, next)) return false;
                    any_scan = true;
                    break;
                default: break;  // other application / comment segments
            }
            p = next;
        }
        if (!frame_seen || !any_scan) return fail("no image data");
        return true;
    }

    void assemble(std::vector<unsigned char>& rgb) {
        rgb.assign((size_t)width * height * 3, 0);
        const size_t n = comps.size();
        std::vector<RowWalker> walkers;
        std::vector<std::vector<uint8_t>> lines(n);
        for (size_t k = 0; k < n; ++k) {
            walkers.emplace_back(v_max / comps[k].v, comps[k].height);
            lines[k].resize((size_t)width + 8 * (size_t)h_max);
        }
        // components named R, G, B, or an Adobe file without colour transform: already RGB
        // (stb_image.h:3879)
        bool is_rgb = false;
        if (n == 3) {
            const bool named = comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B';
            is_rgb = named || (adobe_transform == 0 && !jfif);
        }
        for (int j = 0; j < height; ++j) {
            for (size_t k = 0; k < n; ++k) {
                const Component& c = comps[k];
                const int hs = h_max / c.h, vs = v_max / c.v;
                int nr, fr;
                walkers[k].current(nr, fr);
                upsample_row(&c.plane[(size_t)nr * c.stride], &c.plane[(size_t)fr * c.stride], (width + hs - 1) / hs, hs, vs,
                             lines[k].data());
                walkers[k].advance();
            }
            uint8_t* out = &rgb[(size_t)j * width * 3];
            if (n == 1) {
                for (int i = 0; i < width; ++i) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = lines[0][(size_t)i];
            } else if (is_rgb) {
                for (int i = 0; i < width; ++i)
                    for (size_t k = 0; k < 3; ++k) out[3 * i + (int)k] = lines[k][(size_t)i];
            } else {
                ycc_to_rgb(lines[0].data(), lines[1].data(), lines[2].data(), width, out);
            }
        }
    }
};

}  // namespace

bool load_jpeg(const std::vector<unsigned char>& data, int& w, int& h, std::vector<unsigned char>& rgb, std::string& why) {
    Decoder dec(data, why);
    if (!dec.parse()) return false;
    w = dec.width;
    h = dec.height;
    dec.assemble(rgb);
    return true;
}

}  // namespace detail
}  // namespace rt
