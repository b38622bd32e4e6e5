// drt_fused.hip -- BASELINE config 5: the `nerf` emission-absorption march (python/integrators/nerf.py:47-165,
// 128 queries per ray, opt_config.py:162-169) FUSED with differential-ratio-tracking scattering
// (python/integrators/volpathsimple.py:38-655) in ONE pass over ONE interleaved four-channel grid.
//
// The reference's scenes bind the same asset as the medium's albedo and emission grid
// (python/scene_config.py:109-110: albedo_filename == emission_filename), so sigma_t and ONE colour grid describe
// the medium for both integrators.  The library keeps an interleaved [sigma_t, r, g, b] copy of them with 16-byte
// voxels in an apron-brick layout (Params::grid4, eval4 in drt_device.h: every trilinear footprint of all four
// channels in two 128-byte lines).  Per ray, one kernel
//   1. marches the nerf queries (nerf.py:94-129): ONE four-channel lookup per query gives sigma_t and the emission
//      at the same point (nerf.py:151-165); adjoint: ONE 32-byte record per query carries d/d sigma_t and
//      d/d emission (drt_deferred.hip stream 1) - a 16-byte sigma_t record where the query has no weight
//      (empty space), nothing where both vanish;
//   2. traces the volpathsimple path with the wave-cooperative tracker (drt_coop_tracer.h) from the same camera
//      ray; sigma_t and albedo at scatter points come from the same four-channel copy.
// Each half is bit-identical to its stand-alone integrator on the same ray (same PCG32 stream tea32(seed, index),
// same arithmetic), so the fused pass equals NeRFIntegrator.sample + VolpathSimpleIntegrator.sample; the
// gradients of both land in ONE pair of grids: d/d sigma_t (Z,Y,X,1) and d/d colour (Z,Y,X,3) = albedo gradient +
// emission gradient (the grids are one parameter).
#include "drt_fused_kernel.h"
#include "drt_launch.h"

namespace drt {

namespace {

// caller's sigma_t (Z,Y,X,1) + colour (Z,Y,X,3) -> interleaved four-channel apron-brick copy (see eval4);
// one thread per stored float4
__global__ void __launch_bounds__(256) brick_grid4_kernel(const float *sigma_t, const float *rgb, float4 *dst, int rx, int ry,
                                                          int rz, int nbx)
{
    const size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t) nbx * ry * rz * 16;
    if (t >= total) return;
    const uint32_t slot = (uint32_t) (t & 15), line = (uint32_t) (t >> 4);
    const uint32_t bx = line % (uint32_t) nbx, r = line / (uint32_t) nbx;
    const uint32_t y0 = r % (uint32_t) ry, z0 = r / (uint32_t) ry;
    const int x = min((int) (3 * bx + (slot & 3)), rx - 1);
    const int y = min((int) (y0 + ((slot >> 2) & 1)), ry - 1);
    const int z = min((int) (z0 + (slot >> 3)), rz - 1);
    const size_t v = ((size_t) z * ry + y) * rx + x;
    dst[t] = make_float4(sigma_t[v], rgb[3 * v], rgb[3 * v + 1], rgb[3 * v + 2]);
}

}  // namespace

hipError_t launch_brick_grid4(const float *sigma_t, const float *rgb, float4 *dst, int rx, int ry, int rz, int nbx,
                              hipStream_t stream)
{
    const size_t total = (size_t) nbx * ry * rz * 16;
    hipLaunchKernelGGL(brick_grid4_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream, sigma_t, rgb, dst, rx, ry, rz, nbx);
    return hipGetLastError();
}

// constant emitter, global majorant here; the other three pairs in their own translation units
hipError_t launch_fused(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    const bool env = P.env_pix != nullptr, super = P.mgrid != nullptr;
    if (env && super) return launch_fused_env_super(P, adjoint, count, stream);
    if (env) return launch_fused_env(P, adjoint, count, stream);
    if (super) return launch_fused_super(P, adjoint, count, stream);
    return launch_fused_t<false, false>(P, adjoint, count, stream);
}

}  // namespace drt
