import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import uivr_amd as uivr
gpu = torch.device("cuda", 0)
rng = np.random.default_rng(11)
res = (20, 16, 24)
st = rng.random((res[2], res[1], res[0], 1), dtype=np.float32) * 8.0
st[rng.random(st.shape) < 0.6] = 0.0
st[:, :, :8] = 0.0
al = (rng.random((res[2], res[1], res[0], 3), dtype=np.float32) * 0.8 + 0.1).astype(np.float32)
medium = uivr.GridMedium(sigma_t=st, albedo=al, bbox_min=(-1, -1, -1), bbox_max=(1, 0.8, 1.4), scale=1.3, majorant_resolution_factor=2)
sensor = uivr.PerspectiveSensor(origin=(3.0, 2.0, 4.0), target=(0, 0, 0.2), fov=35.0, width=24, height=24)
scene = uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter((0.9, 1.0, 1.1)), sensors=[sensor])
spp, seed = 8, 31
sg = uivr.scene_to(scene, gpu)
integ = uivr.get_int_config("volpathsimple-drt").create(max_depth=64)
h = integ.native_handle(sg)
h.enable_counters(True); h.reset_counters()
img = uivr.render_primal(sg, integ, 0, spp, seed)
print("primal", {k: int(v) for k, v in h.get_counters().items()})
h.reset_counters()
grads = uivr.render_backward(sg, integ, ((2.0 / img.numel()) * (img - 0.5)).contiguous(), 0, spp, seed)
torch.cuda.synchronize()
print("backward", {k: int(v) for k, v in h.get_counters().items()})
