"""MI355X-native differential-ratio-tracking (DRT) volume integrator.

Host-side mirror of the reference's plugin surface for the hot path
(`VolpathSimpleIntegrator.sample`, python/integrators/volpathsimple.py) over a
hand-written HIP library (csrc/, C ABI in include/drt_hip.h).  Importing this
package does not need a GPU; creating an integrator handle does.
"""
from .scene import (ALBEDO_KEY, EMISSION_KEY, SIGMA_T_KEY, ConstantEmitter, EnvmapEmitter, GridMedium,
                    PerspectiveSensor, Scene, cube_test_scene, scene_to)
from .integrators import (ADMode, FusedNerfDrtIntegrator, IndependentSampler, NeRFIntegrator, RayBatch, VolpathSimpleIntegrator, load_dict,
                          register_integrator, sample_tea_32)
from .opt_config import IntegratorConfig, add_int_config, get_int_config
from .distributed import (GradientSupport, ShardSpec, allreduce_gradients, allreduce_scalar, from_environment, gradient_support,
                          local_loss_scale, reset_allreduce_state, verify_pending)
from .render import alloc_grads, render, render_backward, render_primal
from .batched import gather_ref_values, render_batch, sample_batch, sensors_to_device
from . import losses
from .optimize import (Adam, OptimizationConfig, SGD, SceneConfig, Schedule, adjusted_majorant_res_factor,
                       enforce_valid_params, get_reference_image_paths, load_reference_images, render_previews,
                       render_reference_image, run_optimization, save_params, upsample_grid)
from .volume_io import medium_from_vol, read_vol, write_vol
from .image_io import read_image, write_image
from .fd import fd_gradients

__all__ = [
    "ALBEDO_KEY", "EMISSION_KEY", "SIGMA_T_KEY", "ConstantEmitter", "EnvmapEmitter", "GridMedium", "PerspectiveSensor",
    "Scene", "cube_test_scene", "scene_to", "ADMode", "IndependentSampler", "RayBatch",
    "VolpathSimpleIntegrator", "NeRFIntegrator", "FusedNerfDrtIntegrator", "load_dict", "register_integrator", "sample_tea_32", "IntegratorConfig",
    "add_int_config", "get_int_config", "ShardSpec", "allreduce_gradients", "allreduce_scalar", "GradientSupport", "gradient_support",
    "reset_allreduce_state", "verify_pending",
    "from_environment", "local_loss_scale", "alloc_grads", "render", "render_backward", "render_primal", "render_batch",
    "gather_ref_values", "sample_batch", "sensors_to_device", "losses", "Adam", "SGD", "OptimizationConfig",
    "SceneConfig", "Schedule", "adjusted_majorant_res_factor", "enforce_valid_params", "run_optimization",
    "save_params", "upsample_grid", "read_vol", "write_vol", "medium_from_vol", "read_image", "write_image", "get_reference_image_paths",
    "load_reference_images", "render_previews", "render_reference_image", "fd_gradients",
]
