// drt_coop_super.hip -- the tracer kernels for scenes with a majorant supergrid (majorant_resolution_factor > 0,
// the reference's default: python/scene_config.py:36): CoopTracer with SUPER, i.e. the same bounce loop, path cache,
// record streams and estimator specialisation as the global-majorant kernels, every tracking step on its own lane.
#include "drt_coop_kernel.h"

namespace drt {

hipError_t launch_trace_coop_super(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    return launch_trace_coop_t<true>(P, adjoint, count, stream);
}

}  // namespace drt
