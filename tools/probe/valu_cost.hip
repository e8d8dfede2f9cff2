// Issue cost per instruction kind at full occupancy (8 waves / SIMD, 8 independent chains per lane),
// reported relative to v_fma_f32.
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
            if (OP == 1) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[k]), "v"(c1) : "vcc");
            if (OP == 2) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(f[k]), "v"(c1) : "s20", "s21");
            if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[k]) : "v"(c1) : );
            if (OP == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(f[k]) : "v"(c1));
            if (OP == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 6) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[k]) : "v"(m));
            if (OP == 7) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(f[k]));
            if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
            if (OP == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(f[k]) : "v"(c2) : );
            if (OP == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*reinterpret_cast<uint64_t*>(&a[k & 6])) : "v"(m), "v"(a[7]) : "vcc");
            if (OP == 12) asm volatile("v_sin_f32 %0, %0" : "+v"(f[k]));
            if (OP == 13) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 14) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(f[k]) : "v"(f[(k + 1) & 7]), "v"(c1) : "vcc");
            if (OP == 15) asm volatile("v_cmp_le_f32_e64 s[20:21], s22, %0" : : "v"(f[k]) : "s20", "s21");
            if (OP == 16) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
            if (OP == 17) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
            if (OP == 18) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
            if (OP == 19) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
            if (OP == 20) asm volatile("v_sub_f32_e64 %0, |%0|, %1" : "+v"(f[k]) : "v"(c1));
            if (OP == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f[k]) : "v"(c1));
            if (OP == 22) asm volatile("v_cmp_le_f32_e64 s[20:21], |%0|, %1" : : "v"(f[k]), "v"(c1) : "s20", "s21");
            if (OP == 23) asm volatile("v_mul_f32 %0, s22, %0" : "+v"(f[k]));
            if (OP == 24) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 25) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[k]), "v"(m) : "vcc");
            if (OP == 26) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 27) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
            if (OP == 28) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(f[k]));
            if (OP == 29) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(a[7]));
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
    printf("%-34s %8.3f ms  %5.2f cycles @2.4GHz  %.2f x fma\n", name, best, cyc, best / g_fma);
    return 0;
}

int main() {
    uint32_t* d;
    CHK(hipMalloc(&d, 256 * 8 * 256 * 4));
    run<0>("v_fma_f32", d);
    run<1>("v_cmp_lt_f32 vcc", d);
    run<2>("v_cmp_lt_f32 sgpr pair", d);
    run<3>("v_cndmask_b32 (vcc)", d);
    run<4>("v_mov_b32", d);
    run<5>("v_mul_lo_u32", d);
    run<6>("v_lshrrev_b32", d);
    run<7>("v_cvt_f32_u32", d);
    run<8>("v_rcp_f32", d);
    run<9>("v_mul_f32", d);
    run<10>("v_fma_f32 with sgpr operand", d);
    run<11>("v_mad_u64_u32", d);
    run<12>("v_sin_f32", d);
    run<13>("v_xor_b32", d);
    run<14>("v_cmp + dependent v_cndmask", d);
    run<15>("v_cmp_le_f32_e64 sdst, sgpr, vgpr", d);
    run<16>("v_fmac_f32 (VOP2)", d);
    run<17>("v_min_f32", d);
    run<18>("v_max3_f32", d);
    run<19>("v_med3_f32", d);
    run<20>("v_sub_f32 |abs|", d);
    run<21>("v_cndmask_b32_e64 sgpr mask", d);
    run<22>("v_cmp_le_f32_e64 |abs|", d);
    run<23>("v_mul_f32 sgpr (VOP2)", d);
    run<24>("v_add_u32", d);
    run<25>("v_cmp_lt_u32 vcc", d);
    run<26>("v_and_or_b32", d);
    run<27>("v_sqrt_f32", d);
    run<28>("v_cvt_u32_f32", d);
    run<29>("v_mad_u32_u24", d);
    return 0;
}
