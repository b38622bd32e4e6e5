// drt_launch.h -- host-side launch interface between the C ABI (drt_capi.cpp)
// and the kernels (drt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "drt_device.h"

namespace drt {

hipError_t launch_trace(const Params &P, bool adjoint, bool count, hipStream_t stream);
hipError_t launch_majorant_grid(const float *sigma_t, int rx, int ry, int rz, int gx, int gy, int gz, float scale,
                                float *out, hipStream_t stream);
hipError_t launch_occupancy(const float *sigma_t, int rx, int ry, int rz, int shift, int ox, int oy, int oz,
                            uint32_t *occ, int words, hipStream_t stream);
hipError_t launch_brick_sigma(const float *src, float *dst, int rx, int ry, int rz, int nbx, int nby,
                              hipStream_t stream);
hipError_t launch_nerf(const Params &P, bool adjoint, bool count, hipStream_t stream);
hipError_t launch_trace_wavefront(const Params &P, bool adjoint, bool count, int n_cus, hipStream_t stream);
hipError_t launch_untile(const Params &P, hipStream_t stream);
hipError_t launch_majorant(const float *sigma_t, size_t n, float scale, uint32_t *scratch_bits,
                           float *majorant, hipStream_t stream);
hipError_t launch_batch_raygen(const float *sensors, int n_sensors, uint32_t batch_size, uint32_t spp, uint32_t seed_pixels,
                               uint32_t seed_rays, float *rays_o, float *rays_d, uint32_t *sensor_idx, uint32_t *pixels,
                               hipStream_t stream);
hipError_t launch_film_develop(const float *L, uint64_t n_pixels, uint32_t spp, float *image,
                               hipStream_t stream);
hipError_t launch_film_backward(const float *grad_image, uint64_t n_pixels, uint32_t spp, float *dL,
                                hipStream_t stream);
uint32_t host_alt_seed(uint32_t seed, bool sensor_flow);
hipError_t launch_debug_eval(const Params &P, int op, const float *in, uint64_t n, float *out, hipStream_t stream);

}  // namespace drt
