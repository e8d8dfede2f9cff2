// rtx_launch.h — host-callable launchers exported by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "rtx_device.h"

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
