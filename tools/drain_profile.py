"""When do the supergrid tracer's waves run out of rays, and when do they end?  Needs an experiment build of drt_super.hip:

    tools/mk_variant.sh drain "-DDRT_SUPER_PROFILE=3"; LD_LIBRARY_PATH=variants/drain python tools/drain_profile.py

    queued tracer (drt_sq.hip): tools/mk_variant.sh drain5 "-DDRT_SQ_PROFILE=5" drt_sq.hip; LD_LIBRARY_PATH=variants/drain5 python tools/drain_profile.py

(headline scene at majorant_resolution_factor 8; times in microseconds from the first wave's start)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uivr_amd as u
from uivr_amd import synthetic

dev = torch.device("cuda", 0)
scene = synthetic.dust_devil_scene(res=256, film=512, device=dev)
scene.medium.majorant_resolution_factor = 8
spp = int(os.environ.get("DRT_PROFILE_SPP", "32"))
sensor = scene.sensors[0]
integ = u.get_int_config("volpathsimple-drt").create(max_depth=64)
batch = u.RayBatch(n_rays=sensor.width * sensor.height * spp, spp=spp, sensor=sensor, ray_offset=0, interleave=None)
grads = u.alloc_grads(scene)
h = integ.native_handle(scene)
sampler = u.IndependentSampler(u.sample_tea_32(7, 988378)[0], spp)
L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)
h.enable_counters(True)
h.reset_counters()
L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)
cp = list(h.get_counters().values())
img = integ.develop(scene, L, spp)
dL = integ.film_backward(scene, (2.0 / (img.numel())) * (img - 0.5), spp)
h.reset_counters()
integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)
ca = list(h.get_counters().values())
q62 = 1 << 62
for tag, c in (("primal", cp), ("adjoint", ca)):
    c = [int(v) for v in c]
    t0 = q62 - c[0]
    waves = max(1, c[4])
    print(f"{tag}: waves {waves}; first wave ends {(q62 - c[1] - t0) / 100:.0f} us, mean end {c[3] / waves / 100:.0f} us, LAST end {(c[2] - t0) / 100:.0f} us; "
          f"queues dry: first {(q62 - c[6] - t0) / 100:.0f} us, mean {c[5] / waves / 100:.0f} us")
