// drt_fused_env_super.hip -- the fused nerf + volpathsimple pass (drt_fused.hip, drt_fused_kernel.h): environment-map emitter and majorant supergrid: the set-up of the reference's nerf scenes (python/scene_config.py:36,102-141).
#include "drt_fused_kernel.h"
#include "drt_launch.h"

namespace drt {

hipError_t launch_fused_env_super(const Params &P, bool adjoint, bool count, hipStream_t stream)
{
    return launch_fused_t<true, true>(P, adjoint, count, stream);
}

}  // namespace drt
