import sys, time, os
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import uivr_amd as u
from uivr_amd import synthetic
from oracle import binding as ob
print("cores", os.cpu_count())
sc = synthetic.dust_devil_scene(res=256, film=512, device="cpu")
cpu_scene = u.Scene(medium=u.GridMedium(sigma_t=sc.medium.sigma_t.numpy(), albedo=sc.medium.albedo.numpy(), bbox_min=sc.medium.bbox_min, bbox_max=sc.medium.bbox_max, scale=sc.medium.scale, majorant_resolution_factor=8), emitter=sc.emitter, sensors=sc.sensors)
osc = ob.OracleScene(cpu_scene)
props = u.get_int_config("volpathsimple-drt").create(max_depth=64).props()
ob.render_primal(osc, props, 1, 7)
for spp in (8, 32):
    t=time.perf_counter(); L,_=ob.render_primal(osc, props, spp, 7); dp=time.perf_counter()-t
    t=time.perf_counter(); r=ob.h1_step(osc, props, spp, 7); dt=time.perf_counter()-t
    t=time.perf_counter(); r=ob.h1_step(osc, props, spp, 7, grad_cache_log2=16); dc=time.perf_counter()-t
    print(spp, "primal only", round(dp,2), "h1", round(dt,2), "h1 cached", round(dc,2), round(512*512*spp/dt/1e6,3), "Msamples/s", flush=True)
t=time.perf_counter(); a=np.zeros((256,256,256,4),dtype=np.float64); a[...]=1.0; print("touch 537MB", round(time.perf_counter()-t,2))
