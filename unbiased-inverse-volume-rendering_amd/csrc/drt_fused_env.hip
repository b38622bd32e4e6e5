// drt_fused_env.hip -- the fused nerf + volpathsimple pass (drt_fused.hip, drt_fused_kernel.h): environment-map emitter, global majorant.
#include "drt_fused_kernel.h"
#include "drt_launch.h"

namespace drt {

hipError_t launch_fused_env(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    return launch_fused_t<true, false>(P, adjoint, count, stream);
}

}  // namespace drt
