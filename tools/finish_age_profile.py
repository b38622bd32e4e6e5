"""When do the queued supergrid tracer's PATHS end, how old are they then, and how long were they?  Needs an experiment build:

    tools/mk_variant.sh prof6 "-DDRT_SQ_PROFILE=6" drt_sq.hip
    LD_LIBRARY_PATH=variants/prof6 DRT_PROFILE_SPP=32 python tools/finish_age_profile.py

Per 0.25-ms bucket of the workgroups' clocks (time since the workgroup's start): paths that ended there, their mean / largest age (time
since their ray was started), their mean number of bounce-loop iterations (main path + recursive path), and the share of them older than
half of the launch so far - i.e. whether the launch's last paths are OLD paths slowed down by queueing behind younger ones (priority
would help) or paths that started late (only the start order helps).  Headline scene at majorant_resolution_factor 8.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uivr_amd as u
from uivr_amd import synthetic

dev = torch.device("cuda", 0)
scene = synthetic.dust_devil_scene(res=256, film=512, device=dev)
scene.medium.majorant_resolution_factor = int(os.environ.get("FACTOR", "8"))
spp = int(os.environ.get("DRT_PROFILE_SPP", "32"))
sensor = scene.sensors[0]
integ = u.get_int_config("volpathsimple-drt").create(max_depth=64)
batch = u.RayBatch(n_rays=sensor.width * sensor.height * spp, spp=spp, sensor=sensor, ray_offset=0, interleave=None)
grads = u.alloc_grads(scene)
h = integ.native_handle(scene)
lib = ctypes.CDLL("libdrt_hip.so")
buf = (ctypes.c_ulonglong * 160)()


def read(reset=True):
    torch.cuda.synchronize()
    rc = lib.drt_sq_debug_read(buf, 160, 1 if reset else 0)
    assert rc == 0, rc
    return [int(v) for v in buf]


def show(tag, v):
    print(f"--- {tag}: paths by the time they END (0.25 ms buckets of the workgroup clock)")
    print("  t_end ms |    paths | mean age ms | max age ms | mean iterations | share older than half the launch")
    tot = sum(v[5 * b] for b in range(32))
    for b in range(32):
        n, age, its, old, mx = v[5 * b: 5 * b + 5]
        if not n:
            continue
        print(f"  {0.25 * b:5.2f}-{0.25 * (b + 1):4.2f} | {n:8d} | {age * 1.28e-3 / n:11.3f} | {mx * 1.28e-3:10.3f} | {its / n:15.2f} | {old / n:6.3f}   ({100.0 * n / tot:5.2f} % of paths)")


sampler = u.IndependentSampler(u.sample_tea_32(7, 988378)[0], spp)
L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)      # warm-up (no counters: nothing recorded)
h.enable_counters(True)
read()
L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)
show(f"primal, {spp} spp", read())
img = integ.develop(scene, L, spp)
dL = integ.film_backward(scene, (2.0 / (img.numel())) * (img - 0.5), spp)
integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)
show(f"adjoint, {spp} spp", read())
