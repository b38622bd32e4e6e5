"""Raw profile slots of an experiment build of the supergrid tracer (drt_super.hip, -DDRT_SUPER_PROFILE=1|2|4):

    tools/mk_variant.sh prof4 "-DDRT_SUPER_PROFILE=4"; LD_LIBRARY_PATH=variants/prof4 python tools/super_profile.py

(drt_sq.hip: tools/mk_variant.sh sqprof "-DDRT_SQ_PROFILE=1" drt_sq.hip; DRT_PROFILE_MODE=sq)
(headline scene at majorant_resolution_factor 8; the meaning of the slots is next to DRT_PROF / DRT_PROF4 / DRT_STAMP in the source)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uivr_amd as u
from uivr_amd import synthetic

dev = torch.device("cuda", 0)
if os.environ.get("DRT_PROFILE_SCENE", "dust") == "smoke":
    scene = synthetic.smoke_scene(res=128, film=512, device=dev)
else:
    scene = synthetic.dust_devil_scene(res=256, film=512, device=dev)
scene.medium.majorant_resolution_factor = int(os.environ.get("DRT_PROFILE_FACTOR", "8"))
if os.environ.get("DRT_PROFILE_ENV"):                    # lit by a 2048x1024 environment map (bench.py: headline_envmap_factor8)
    g = torch.Generator().manual_seed(5)
    scene.emitter = u.EnvmapEmitter(pixels=(torch.rand(1024, 2048, 3, generator=g) ** 4 * 3.0 + 0.2).to(dev), scale=1.0)
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
cp = [int(v) for v in h.get_counters().values()]
img = integ.develop(scene, L, spp)
dL = integ.film_backward(scene, (2.0 / (img.numel())) * (img - 0.5), spp)
h.reset_counters()
integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)
ca = [int(v) for v in h.get_counters().values()]
print("primal ", cp)
print("adjoint", ca)
mode = os.environ.get("DRT_PROFILE_MODE", "4")
for tag, c in (("primal", cp), ("adjoint", ca)):
    if mode == "sq":
        print(f"{tag}: lane steps {c[0]/1e6:.1f} M, wave steps {c[1]/1e6:.2f} M -> {c[0]/max(1,c[1]):.1f} lanes per step; collision batches {c[2]/1e6:.3f} M x {c[3]/max(1,c[2]):.1f} rays; "
              f"transition batches {c[4]/1e6:.3f} M x {c[5]/max(1,c[4]):.1f}; regeneration batches {c[6]/1e6:.3f} M x {c[7]/max(1,c[6]):.1f}; polls {c[8]/1e6:.2f} M")
    elif mode in ("sq3", "sq4"):
        names = ["DRT vertex", "NEE walk finished", "phase sampling", "loop head", "collision / escape", "emitter direction", "end of path", "flight set-up", "passes"]
        print(tag, "waves" if mode == "sq3" else "lanes", {n: round(v / 1e6, 3) for n, v in zip(names, c)})
    elif mode == "sq2":
        tot = sum(c)
        names = ["cell steps", "walker queue work", "load (collision)", "collision code", "load (transition)", "transition code", "flight set-up", "store + push", "looking for work"]
        print(tag, {n: round(v / tot, 3) for n, v in zip(names, c)})
    elif mode == "4":
        print(f"{tag}: lane steps {c[0]/1e6:.1f} M, wave steps {c[1]/1e6:.2f} M -> {c[0]/max(1,c[1]):.1f} lanes per step; batches {c[2]/1e6:.2f} M, "
              f"{c[3]/max(1,c[2]):.1f} flying at batch start, {c[4]/max(1,c[2]):.1f} finish per batch; transition passes {c[6]/1e6:.2f} M with "
              f"{c[5]/max(1,c[6]):.1f} lanes; regeneration blocks {c[7]/1e6:.3f} M, {c[8]/max(1,c[7]):.1f} rays each")
    elif mode == "2":
        tot = sum(c[k] for k in (1, 3, 4, 5, 6, 7, 8))
        names = {1: "poll/sleep", 3: "epilogue", 4: "regen", 5: "transitions", 8: "set-up", 6: "pull", 7: "steps+writeback"}
        print(tag, {names[k]: round(c[k] / tot, 3) for k in names}, "heavy runs", c[2])
    else:
        print(f"{tag}: cell steps {c[0]/1e6:.1f} M, polls {c[1]/1e6:.2f} M, heavy runs {c[2]/1e6:.2f} M with {c[3]/max(1,c[2]):.1f} lanes ready, "
              f"epilogue lanes {c[4]/max(1,c[2]):.1f}, posted {c[5]/max(1,c[2]):.1f}; pulls {c[6]/1e6:.2f} M x {c[7]/max(1,c[6]):.1f} flights; transition passes {c[8]/1e6:.2f} M")
