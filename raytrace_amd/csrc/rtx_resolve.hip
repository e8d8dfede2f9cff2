// Per-pixel epilogue of the reference's render loop on the device:
// c /= ns (Scene.cpp:150), DenanFilter (DenanFilter.h:8-13), GammaFilter
// (GammaFilter.h:7-9 -> Util.h:41-47, powf(c, 1/2.2)), TonemapFilter
// (TonemapFilter.h:8-10, clamp to [0,1]), then c * 255.99f truncated to 8 bits
// (Image.h:36-38).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "rtx_launch.h"

namespace rtx_post {

__device__ __forceinline__ float clamp01(float c) {
    // maxPerElem(c, 0) then minPerElem(., 1) (vec_aos.h:418-442): a NaN compares false
    // and yields 0
    float m = (c > 0.0f) ? c : 0.0f;
    return (m < 1.0f) ? m : 1.0f;
}

__global__ void __launch_bounds__(256) resolve_kernel(const float* __restrict__ accum,
                                                      uint8_t* __restrict__ rgb8, uint32_t n_pixels,
                                                      float ns) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pixels) return;
    float r = accum[3u * k + 0u] / ns;
    float g = accum[3u * k + 1u] / ns;
    float b = accum[3u * k + 2u] / ns;
    r = !(r == r) ? 1.0f : r;
    g = !(g == g) ? 0.0f : g;
    b = !(b == b) ? 1.0f : b;
    const float inv_gamma = 1.0f / 2.2f;
    r = clamp01(powf(r, inv_gamma));
    g = clamp01(powf(g, inv_gamma));
    b = clamp01(powf(b, inv_gamma));
    rgb8[3u * k + 0u] = (uint8_t)(r * 255.99f);
    rgb8[3u * k + 1u] = (uint8_t)(g * 255.99f);
    rgb8[3u * k + 2u] = (uint8_t)(b * 255.99f);
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
