// drt_own.hip -- the kernels for scenes whose COLOUR grids (albedo; the nerf integrator's emission) live on their own lattice
// (drt_set_colour_resolution): Mitsuba interpolates every GridVolume on its own resolution, and the reference's janga-smoke pairs a
// 264 x 136 x 136 density with 256 x 128 x 128 albedo / emission grids (python/scene_config.py:108-110).
//
// The ONE translation unit compiled with DRT_COLOUR_OWN: eval_rgb builds its stencil from Params::crx / cry / crz and the colour gradient
// splats go to the caller's grid as fp32 atomics on that lattice (drt_device.h: make_stencil_colour, splat_colour_own), because the tile
// partition, the LDS reduction tiles and the apron scratch are laid out on sigma_t's lattice.  Everything else - the wave-cooperative tracer
// with a global majorant, its own-lane walk through a majorant supergrid, path cache, sigma_t record streams, every estimator, both emitters,
// the nerf march - is the code of the other units, instantiated here once more (internal linkage).  A scene with equal lattices never gets
// here: its kernels, and the benchmark, are untouched by this unit.  Parity: tests/test_gpu_lattice.py against the oracle.
#define DRT_COLOUR_OWN 1
#include "drt_coop_kernel.h"
#include "drt_nerf_kernel.h"

namespace drt {

hipError_t launch_trace_own(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    if (P.mgrid) return launch_trace_coop_t<true>(P, adjoint, count, stream);
    return launch_trace_coop_t<false>(P, adjoint, count, stream);
}

hipError_t launch_nerf_own(const Params &P, bool adjoint, bool count, hipStream_t stream) { return launch_nerf_t(P, adjoint, count, stream); }

}  // namespace drt
