// Issue cost of further instruction kinds at 8 waves / SIMD, 8 independent chains per lane (see valu_cost.hip):
// integer min / max, v_cmpx, SDWA operand selection, packed fp32, VOP2 with a scalar operand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 1 << 15;

template <int OP>
__global__ void __launch_bounds__(256) probe(uint32_t* out, uint32_t seed, float one) {
    float f[8];
    uint32_t a[8];
    for (int k = 0; k < 8; ++k) { a[k] = seed + threadIdx.x + k; f[k] = 1.0f + (float)(a[k] & 1023u) * 1e-3f; }
    const float c1 = one * 1.0000001f, c2 = one * 1e-7f;
    const uint32_t m = seed * 747796405u | 1u;
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
            if (OP == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 2) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 3) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 4) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 5) asm volatile("v_cmpx_le_f32 vcc, %1, %0" : : "v"(f[k]), "v"(c2) : "vcc");  // always true: exec stays full
            if (OP == 6) asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(f[k]) : "v"(a[k]), "v"(c1));
            if (OP == 7) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(a[k]) : "v"(a[(k + 1) & 7]), "v"(m));
            if (OP == 8) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[k]) : "v"(a[k]));
            if (OP == 9) asm volatile("v_fmac_f32 %0, s22, %1" : "+v"(f[k]) : "v"(c2));
            if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<uint64_t*>(&f[k & 6])) : "v"(*reinterpret_cast<const uint64_t*>(&f[(k + 2) & 6])), "v"(*reinterpret_cast<const uint64_t*>(&f[(k + 4) & 6])));
            if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, s[22:23], %1" : "+v"(*reinterpret_cast<uint64_t*>(&f[k & 6])) : "v"(*reinterpret_cast<const uint64_t*>(&f[(k + 4) & 6])));
            if (OP == 12) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<uint64_t*>(&f[k & 6])) : "v"(*reinterpret_cast<const uint64_t*>(&f[(k + 2) & 6])));
            if (OP == 13) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 14) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x48" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 15) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[k]) : : "vcc");
            if (OP == 16) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 17) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(f[k]) : "v"(c2));
            if (OP == 18) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800001" : "+v"(f[k]) : "v"(c1));
            if (OP == 19) asm volatile("v_add_f32 %0, s22, %0" : "+v"(f[k]));
            if (OP == 20) asm volatile("v_mov_b32 %0, s22" : "=v"(f[k]));
            if (OP == 21) asm volatile("v_cmp_le_f32 vcc, s22, %0" : : "v"(f[k]) : "vcc");
            if (OP == 22) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(f[k]) : "v"(f[(k + 1) & 7]), "v"(c1));
            if (OP == 23) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[k]) : "v"(a[k]));
            if (OP == 24) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 25) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
        }
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; ++k) r ^= a[k] ^ __float_as_uint(f[k]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static double g_fma = 0;
template <int OP>
int run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    probe<OP><<<blocks, 256>>>(d, 1, 1.0f);
    CHK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        probe<OP><<<blocks, 256>>>(d, 2, 1.0f);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    if (OP == 0) g_fma = best;
    double cyc = best * 1e-3 * 2.4e9 / ((double)kIters * 16 * 8);
    printf("%-40s %8.3f ms  %5.2f cycles @2.4GHz  %.2f x fma\n", name, best, cyc, best / g_fma);
    return 0;
}

int main() {
    uint32_t* d;
    CHK(hipMalloc(&d, 256 * 8 * 256 * 4));
    run<0>("v_fma_f32", d);
    run<1>("v_max_i32", d);
    run<2>("v_min_i32", d);
    run<24>("v_max_u32", d);
    run<3>("v_max3_i32", d);
    run<4>("v_min3_u32", d);
    run<5>("v_cmpx_le_f32", d);
    run<21>("v_cmp_le_f32 vcc, sgpr, vgpr (VOPC e32)", d);
    run<6>("v_mul_f32_sdwa byte operand", d);
    run<7>("v_or_b32_sdwa word operand", d);
    run<22>("v_add_f32_sdwa (plain selects)", d);
    run<8>("v_cvt_f32_ubyte1", d);
    run<23>("v_cvt_f32_u32_sdwa word", d);
    run<9>("v_fmac_f32 sgpr (VOP2)", d);
    run<19>("v_add_f32 sgpr (VOP2)", d);
    run<20>("v_mov_b32 sgpr", d);
    run<10>("v_pk_fma_f32", d);
    run<11>("v_pk_fma_f32 sgpr pair", d);
    run<12>("v_pk_add_f32", d);
    run<13>("v_alignbit_b32", d);
    run<14>("v_bitop3_b32", d);
    run<15>("v_addc_co_u32", d);
    run<16>("v_mul_u32_u24", d);
    run<17>("v_fma_f32 inline constant", d);
    run<18>("v_fmaak_f32 literal", d);
    run<25>("v_sub_f32", d);
    return 0;
}
