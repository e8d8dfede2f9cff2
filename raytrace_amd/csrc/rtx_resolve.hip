// Per-pixel epilogue of the reference's render loop on the device:
// c /= ns (Scene.cpp:150), DenanFilter (DenanFilter.h:8-13), GammaFilter
// (GammaFilter.h:7-9 -> Util.h:41-47, powf(c, 1/2.2)), TonemapFilter
// (TonemapFilter.h:8-10, clamp to [0,1]), then c * 255.99f truncated to 8 bits
// (Image.h:36-38) - byte for byte the reference's output.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "rtx_launch.h"

namespace rtx_post {

// The chain gamma -> clamp -> * 255.99f truncated is a monotone step function of its fp32 input,
// so 255 thresholds describe it completely: kThreshold[k] is the smallest value whose code is
// >= k under the reference's own chain and libm powf (generated from the reference and checked
// against it on every float in [0, 1]; tests/golden/make_golden.py).  Searching the table gives
// the reference's bytes exactly, which no device powf does.  Negative inputs and -0 get code 0
// (powf of a negative finite base is NaN, which the clamp turns into 0; vec_aos.h:418-442),
// except -inf: powf(-inf, 1/2.2) is +inf, code 255.
__constant__ uint32_t kThresholdBits[256] = {
#include "rtx_resolve_table.inc"
};

__device__ __forceinline__ uint32_t code_of(float c, const float* table) {
    c = (c == -__builtin_inff()) ? __builtin_inff() : c;
    uint32_t k = 0;
#pragma unroll
    for (uint32_t step = 128u; step != 0u; step >>= 1) k += (c >= table[k + step]) ? step : 0u;
    return k;
}

__global__ void __launch_bounds__(256) resolve_kernel(const float* __restrict__ accum,
                                                      uint8_t* __restrict__ rgb8, uint32_t n_pixels,
                                                      float ns) {
    __shared__ float s_table[256];
    s_table[threadIdx.x] = __uint_as_float(kThresholdBits[threadIdx.x]);
    __syncthreads();
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pixels) return;
    float r = __fdiv_rn(accum[3u * k + 0u], ns);  // c /= ns: three IEEE divides (Scene.cpp:150)
    float g = __fdiv_rn(accum[3u * k + 1u], ns);
    float b = __fdiv_rn(accum[3u * k + 2u], ns);
    r = !(r == r) ? 1.0f : r;  // DenanFilter.h:8-13
    g = !(g == g) ? 0.0f : g;
    b = !(b == b) ? 1.0f : b;
    rgb8[3u * k + 0u] = (uint8_t)code_of(r, s_table);
    rgb8[3u * k + 1u] = (uint8_t)code_of(g, s_table);
    rgb8[3u * k + 2u] = (uint8_t)code_of(b, s_table);
}

__global__ void __launch_bounds__(256) add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    // grid-stride, one float4 per step where the length allows; HBM bound: 12 bytes per element
    const size_t n4 = n / 4u;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4* d4 = reinterpret_cast<float4*>(dst);
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += stride) {
        float4 a = d4[k];
        const float4 b = s4[k];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        d4[k] = a;
    }
    for (size_t k = 4u * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) dst[k] += src[k];
}

hipError_t launch_add(float* d_dst, const float* d_src, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t blocks = std::min<size_t>((n / 4u + 255u) / 256u + 1u, 256u * 16u);
    hipLaunchKernelGGL(add_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_dst, d_src, n);
    return hipGetLastError();
}

hipError_t launch_resolve(const float* d_accum, uint8_t* d_rgb8, uint32_t n_pixels, uint32_t ns,
                          hipStream_t stream) {
    const dim3 grid((n_pixels + 255u) / 256u), block(256);
    hipLaunchKernelGGL(resolve_kernel, grid, block, 0, stream, d_accum, d_rgb8, n_pixels, (float)ns);
    return hipGetLastError();
}

}  // namespace rtx_post
