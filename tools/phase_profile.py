"""Where the cooperative tracers spend their cycles (headline workload).  Needs an experiment build:

    DRT_PHASE_PROFILE=1 python -c "import __graft_entry__ as g; g.build()"; python tools/phase_profile.py
    python -c "import __graft_entry__ as g; g.build()"        # back to the production build

In that build the event-counter slots of the counting kernels hold shader cycles per phase (summed over waves) instead
of event counts - drt_coop_tracer.h `phase()`.  Cycles are wall time of a wave (stalls and the other waves of its SIMD
included), so the shares are shares of wave residency, which is what bounds a kernel that runs at full occupancy.
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uivr_amd as u
from uivr_amd import synthetic

dev = torch.device("cuda", 0)
scene = synthetic.dust_devil_scene(res=256, film=512, device=dev)
scene.medium.majorant_resolution_factor = int(os.environ.get("DRT_PROFILE_FACTOR", "0"))     # 8: the supergrid kernels
spp = 32
sensor = scene.sensors[0]
integ = u.get_int_config("volpathsimple-drt").create(max_depth=64)
batch = u.RayBatch(n_rays=sensor.width * sensor.height * spp, spp=spp, sensor=sensor, ray_offset=0, interleave=None)
grads = u.alloc_grads(scene)
h = integ.native_handle(scene)
sampler = u.IndependentSampler(u.sample_tea_32(7, 988378)[0], spp)
L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)      # warm-up (path cache sizes, ...)
h.enable_counters(True)
h.reset_counters()
L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)
cp = {k: int(v) for k, v in h.get_counters().items()}
img = integ.develop(scene, L, spp)
dL = integ.film_backward(scene, (2.0 / (img.numel())) * (img - 0.5), spp)
h.reset_counters()
integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)
ca = {k: int(v) for k, v in h.get_counters().items()}
names = {"n_dt": "main path: delta tracking (adjoint: path-cache read)", "n_rt": "main path: NEE (emitter sample + ratio tracking; adjoint: + replay with splats)",
         "n_tr": "main path: everything else (RR, albedo, reservoir, transmittance / scatter splats, next direction, box hit; prologue)",
         "n_drt": "E2 walk (sample_interaction_drt)", "n_alb": "recursive path: delta tracking",
         "n_rt_adj": "recursive path: NEE", "n_sc": "recursive path: everything else"}
mode = int(os.environ.get("DRT_PHASE_PROFILE", "1"))
if mode in (4, 5):
    names = {"n_dt": ">= 33 lanes active", "n_rt": "17..32", "n_drt": "9..16", "n_alb": "5..8", "n_rt_adj": "3..4", "n_sc": "2", "n_sc_alb": "1",
             "n_tr": "everything outside the bounce loop of the " + ("recursive" if mode == 4 else "main") + " path"}
    print("bounce-loop iterations of the", "recursive" if mode == 4 else "main", "path by lanes active at the start of the iteration")
if mode in (2, 3):
    names = {"n_dt": "rounds with >= 33 pending walks (own-lane steps)", "n_rt": "17..32 (m = 2)", "n_drt": "9..16 (m = 4)",
             "n_alb": "5..8 (m = 8)", "n_rt_adj": "3..4 (m = 8)", "n_sc": "2 (m = 8)", "n_sc_alb": "1 (m = 8)",
             "n_tr": "everything outside the " + ("coop_rt" if mode == 2 else "coop_dt") + " rounds"}
for tag, c in (("primal", cp), ("adjoint", ca)):
    tot = sum(c[k] for k in names)
    print(tag, "total wave-cycles (x16/64 units):", tot)
    for k, nm in names.items():
        print(f"   {100.0 * c[k] / max(1, tot):5.1f} %  {nm}")
