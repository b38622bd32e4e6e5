// drt_fused_super.hip -- the fused nerf + volpathsimple pass (drt_fused.hip, drt_fused_kernel.h): constant emitter, majorant supergrid.
#include "drt_fused_kernel.h"
#include "drt_launch.h"

namespace drt {

hipError_t launch_fused_super(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    return launch_fused_t<false, true>(P, adjoint, count, stream);
}

}  // namespace drt
