// How VALU throughput of one SIMD depends on resident waves and on independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 1 << 16;

template <int CHAINS>
__global__ void __launch_bounds__(64) probe(uint32_t* out, uint32_t seed) {
    float f[CHAINS];
    for (int k = 0; k < CHAINS; ++k) f[k] = 1.0f + (float)((seed + threadIdx.x + k) & 1023u) * 1e-3f;
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / CHAINS; ++r)
#pragma unroll
            for (int k = 0; k < CHAINS; ++k)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(1.0000001f), "v"(1e-7f));
    }
    uint32_t r = 0;
    for (int k = 0; k < CHAINS; ++k) r ^= __float_as_uint(f[k]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CHAINS>
int run(int waves_per_simd, uint32_t* d) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 4 * waves_per_simd;  // one wave per block
    // dynamic LDS caps the resident blocks per CU at 4 * waves_per_simd
    const size_t lds = (160 * 1024) / (4 * waves_per_simd) - 512;
    CHK(hipFuncSetAttribute((const void*)probe<CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    probe<CHAINS><<<blocks, 64, lds>>>(d, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    probe<CHAINS><<<blocks, 64, lds>>>(d, 2);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    double instr_per_simd = (double)kIters * 16 * waves_per_simd;
    printf("chains %2d  waves/SIMD %d: %8.3f ms  %.2f cycles per VALU instr per SIMD\n", CHAINS, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
    return 0;
}

int main() {
    uint32_t* d;
    CHK(hipMalloc(&d, 256 * 4 * 8 * 64 * 4));
    for (int w : {1, 2, 4, 8}) {
        run<1>(w, d);
        run<2>(w, d);
        run<4>(w, d);
        run<8>(w, d);
    }
    return 0;
}
