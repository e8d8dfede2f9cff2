// rtx_launch.h — host-callable launchers exported by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "rtx_device.h"

namespace rtxd {

// The pixel tile of one wave.  A wave works through tile pixels x spp items.  Two things pull:
// the lanes of a wave should work on few pixels at a time (neighbouring lanes then meet the same
// surfaces and take the same branches of the integrator), and a wave needs enough items that
// draining its pool - the last paths finish alone - stays a small share of its life.  Measured on
// the Cornell box at 1080p: 4x4 pixels whenever that leaves 2048 items (128 spp and more), else
// the smallest tile that does (8x4 at 64 spp, 8x8 below): 64 spp 81.6 -> 86.7, 128 spp 82.4 ->
// 89.7 Grays/s against the tiles of 8192 items used before; 1024 items (4x4 at 64 spp) 84.5.
#ifndef RTX_TILE_PIXELS_LOG2
#define RTX_TILE_PIXELS_LOG2 4u
#endif
#ifndef RTX_MIN_ITEMS_LOG2
#define RTX_MIN_ITEMS_LOG2 11u
#endif
// ... and at most 8192 items: the kernel ends with its slowest wave, and beyond 512 spp smaller
// tiles (4x2 at 1024 spp, 2x2 from 2048) end 0.3 - 1.6 % sooner than 4x4.
#ifndef RTX_MAX_ITEMS_LOG2
#define RTX_MAX_ITEMS_LOG2 13u
#endif
// BVH scenes: at most 4x4 pixels per wave.  A smaller tile keeps the rays of a wave closer
// together (the lanes fetch the same nodes): 487-sphere field at 64 spp, 8x8 pixels 12.9,
// 4x4 pixels 17.9 Grays/s; at 1024 spp 4x2 pixels (8192 items) 19.2, 2x2 (4096 items) 17.9.
#ifndef RTX_BVH_MAX_TILE_LOG2
#define RTX_BVH_MAX_TILE_LOG2 4u
#endif
// Small frames: whatever the sample count, the launch should have at least about 16 384 waves
// (twice what the 1024 SIMDs hold at eight waves each), else the GPU is mostly empty and the frame
// takes as long as one wave needs for its whole pool.  Measured with 8 192 items per wave against
// tiles that make 20 000 - 40 000 waves: 800 x 400 x 50 spp 1.34 -> 0.70 ms, 400 x 225 x 100 spp
// 3.26 -> 0.61 ms.
#ifndef RTX_MIN_WAVES_LOG2
#define RTX_MIN_WAVES_LOG2 14u
#endif
inline void choose_tile(uint32_t spp, uint32_t width, uint32_t height, bool bvh, uint32_t& w_log2, uint32_t& h_log2) {
    uint32_t spp_log2 = 0;
    while ((1u << spp_log2) < spp && spp_log2 < 31u) ++spp_log2;
    // (but at least 1024 items per wave: at 16 spp the tile is 8x8 again)
    const uint32_t bvh_log2 = spp_log2 + RTX_BVH_MAX_TILE_LOG2 >= 10u ? RTX_BVH_MAX_TILE_LOG2 : 10u - spp_log2;
    const uint32_t max_log2 = bvh ? (bvh_log2 > 6u ? 6u : bvh_log2) : 6u;
    uint32_t pixels_log2 = RTX_TILE_PIXELS_LOG2;
    if (spp_log2 + pixels_log2 < RTX_MIN_ITEMS_LOG2) pixels_log2 = RTX_MIN_ITEMS_LOG2 - spp_log2;
    if (spp_log2 + pixels_log2 > RTX_MAX_ITEMS_LOG2) pixels_log2 = spp_log2 + 2u >= RTX_MAX_ITEMS_LOG2 ? 2u : RTX_MAX_ITEMS_LOG2 - spp_log2;
    if (pixels_log2 > max_log2) pixels_log2 = max_log2;
    // the largest tile that still leaves 2^RTX_MIN_WAVES_LOG2 waves (but not below 2x2 pixels)
    uint32_t frame_log2 = 2u;
    while (frame_log2 < 6u && ((uint64_t)width * height >> (frame_log2 + 1u)) >= (1ull << RTX_MIN_WAVES_LOG2)) ++frame_log2;
    if (pixels_log2 > frame_log2) pixels_log2 = frame_log2;
    w_log2 = (pixels_log2 + 1u) / 2u;
    h_log2 = pixels_log2 / 2u;
}

}  // namespace rtxd

namespace rtx_det_mode {
hipError_t launch_trace(const rtxd::TraceArgs& args, hipStream_t stream);
hipError_t launch_hit(const rtxd::HitArgs& args, hipStream_t stream);
}  // namespace rtx_det_mode

namespace rtx_fast_mode {
hipError_t launch_trace(const rtxd::TraceArgs& args, hipStream_t stream);
hipError_t launch_hit(const rtxd::HitArgs& args, hipStream_t stream);
}  // namespace rtx_fast_mode

namespace rtx_post {
// sum / ns -> NaN filter -> gamma -> clamp -> 8 bit (Scene.cpp:150-151, Image.h:30-39)
hipError_t launch_resolve(const float* d_accum, uint8_t* d_rgb8, uint32_t n_pixels, uint32_t ns,
                          hipStream_t stream);
// dst[k] += src[k]: a partial frame of another GPU added to this one's (fp32, as Scene.cpp:148 adds)
hipError_t launch_add(float* d_dst, const float* d_src, size_t n, hipStream_t stream);
}  // namespace rtx_post
