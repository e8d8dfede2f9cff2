// Issue cost of a few VALU instruction kinds on gfx950, relative to v_fma_f32.
// Each kernel runs 8 independent chains per lane so that dependent-issue latency is hidden.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 1 << 17;

template <int OP>
__global__ void __launch_bounds__(256) probe(uint32_t* out, uint32_t seed) {
    uint32_t a[8];
    float f[8];
    for (int k = 0; k < 8; ++k) { a[k] = seed + threadIdx.x * 8 + k; f[k] = 1.0f + (float)(a[k] & 1023u) * 1e-3f; }
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) f[k] = __builtin_fmaf(f[k], 1.0000001f, 1e-7f);
            if (OP == 1) a[k] = a[k] * 747796405u + 12345u;                      // v_mul_lo_u32 (+ add)
            if (OP == 2) a[k] = (a[k] & 0xffffffu) * 0x5bd1e9u + 7u;  // v_mad_u32_u24
            if (OP == 3) f[k] = __builtin_amdgcn_sqrtf(f[k] + 1.0f);             // v_add + v_sqrt
            if (OP == 4) f[k] = __builtin_amdgcn_rcpf(f[k] + 1.0f);              // v_add + v_rcp
            if (OP == 5) a[k] = (a[k] >> ((a[k] >> 28) + 4u)) ^ a[k];            // shifts / xor
            if (OP == 6) f[k] = f[k] > 1.5f ? f[k] - 0.25f : f[k] + 0.125f;      // cmp + cndmask + ...
            if (OP == 7) f[k] = __builtin_amdgcn_sinf(f[k] * 0.1f);              // v_mul + v_sin
            if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(1.0000001f), "v"(1e-7f));
            if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(747796405u));
            if (OP == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(f[k]) : "v"(1.5f), "v"(0.7f) : "vcc");
            if (OP == 11) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
            if (OP == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
            if (OP == 13) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(0x1234567u));
            if (OP == 14 && (k & 1) == 0)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&f[k])) : "v"(1.0), "v"(2.0));
        }
    }
    uint32_t r = 0;
    for (int k = 0; k < 8; ++k) r ^= a[k] ^ __float_as_uint(f[k]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 4;  // 4 waves per SIMD on 256 CUs
    probe<OP><<<blocks, 256>>>(d, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    probe<OP><<<blocks, 256>>>(d, 2);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    double iters = (double)kIters * 8;  // loop bodies per lane
    // cycles per loop body per wave, assuming 16 waves per CU spread over 4 SIMDs at 2.4 GHz
    double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0);
    printf("%-28s %8.3f ms   %.2f SIMD cycles per body per wave\n", name, ms, cyc);
    return 0;
}

int main() {
    uint32_t* d;
    CHK(hipMalloc(&d, 256 * 4 * 256 * 4));
    run<0>("v_fma_f32", d);
    run<1>("v_mul_lo_u32 + v_add", d);
    run<2>("v_mad_u32_u24", d);
    run<3>("v_add + v_sqrt_f32", d);
    run<4>("v_add + v_rcp_f32", d);
    run<5>("2 shifts + add + xor", d);
    run<6>("cmp + 2 alu + cndmask", d);
    run<7>("v_mul + v_sin_f32", d);
    run<8>("asm v_fma_f32", d);
    run<9>("asm v_mul_lo_u32", d);
    run<10>("asm v_cmp + v_cndmask", d);
    run<11>("asm v_sqrt_f32", d);
    run<12>("asm v_rcp_f32", d);
    run<13>("asm v_xor_b32", d);
    run<14>("asm v_pk_fma_f32 (half count)", d);
    return 0;
}
