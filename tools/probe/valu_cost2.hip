// Issue cost of the instructions a wide-BVH node test is made of (8 waves / SIMD, 8 independent
// chains per lane), relative to v_fma_f32.  Companion of valu_cost.hip.
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
            if (OP == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(f[k]) : "v"(m), "v"(c2));
            if (OP == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(f[k]) : "v"(m), "v"(c2));
            if (OP == 3) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[k]) : "v"(a[k]));
            if (OP == 4) asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(f[k]) : "v"(a[k]));
            if (OP == 5) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[k]) : "v"(a[k]));
            if (OP == 6) asm volatile("v_bfe_u32 %0, %1, 5, 3" : "=v"(a[k]) : "v"(m));
            if (OP == 7) asm volatile("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 8) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
            if (OP == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
            if (OP == 10) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 11) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 12) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 13) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[k]) : "v"(m));
            if (OP == 14) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(a[k]) : "v"(m));
            if (OP == 26) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(m));
            if (OP == 27) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[k]));
            if (OP == 28) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 29) asm volatile("v_cndmask_b32_e64 %0, 0, %1, vcc" : "=v"(a[k]) : "v"(m));
            if (OP == 30) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[k]) : "v"(m));
            if (OP == 31) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(a[k]) : "v"(f[k]), "v"(c1), "v"(m) : "vcc");
            if (OP == 32) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[k]) : "v"(f[k]), "v"(c1), "v"(m) : "vcc");
            if (OP == 15) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 16) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
            if (OP == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 18) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 19) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[k]));
            if (OP == 20) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<uint64_t*>(&a[k & 6])) : "v"(*reinterpret_cast<uint64_t*>(&f[(k + 2) & 6])));
            if (OP == 21) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(m), "v"(a[7]));
            if (OP == 22) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[k]) : "v"(a[k]));
            if (OP == 23) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 24) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (OP == 25) asm volatile("v_mad_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(f[k]) : "v"(m), "v"(c2));
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
    run<1>("v_fma_mix_f32 (f16 lo, f32, f32)", d);
    run<2>("v_fma_mix_f32 (f16 hi, f32, f32)", d);
    run<3>("v_cvt_f32_ubyte0", d);
    run<4>("v_cvt_f32_ubyte2", d);
    run<5>("v_cvt_f32_f16", d);
    run<6>("v_bfe_u32", d);
    run<7>("v_lshl_or_b32", d);
    run<8>("v_min3_f32", d);
    run<9>("v_max_f32", d);
    run<10>("v_pk_fma_f16", d);
    run<11>("v_pk_max_f16", d);
    run<12>("v_pk_min_f16", d);
    run<13>("v_lshlrev_b32", d);
    run<14>("v_cndmask_b32 0, v, vcc", d);
    run<15>("v_or3_b32", d);
    run<16>("v_sub_f32", d);
    run<17>("v_and_b32", d);
    run<18>("v_perm_b32", d);
    run<19>("v_ashrrev_i32", d);
    run<20>("v_pk_mul_f32", d);
    run<21>("v_bfi_b32", d);
    run<22>("v_cvt_f32_i32", d);
    run<23>("v_max_i32", d);
    run<24>("v_min_u32", d);
    run<26>("v_cndmask v, v, v2, vcc (dest=src0)", d);
    run<27>("v_cndmask v, 0, v, vcc (dest=src1)", d);
    run<28>("v_cndmask v, v1, v2, vcc (3 regs)", d);
    run<29>("v_cndmask_e64 v, 0, v1, vcc", d);
    run<30>("v_cndmask v, v1, v, vcc", d);
    run<31>("v_cmp + v_cndmask v,0,v1 (=v)", d);
    run<32>("v_cmp + v_cndmask v,v,v1 (+v)", d);
    return 0;
}
