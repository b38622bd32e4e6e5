"""Cost of distributed.gradient_support (the packing set of the one-collective gradient all-reduce) at 256^3."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import uivr_amd as u
from uivr_amd import synthetic
dev = torch.device('cuda', 0)
scene = synthetic.dust_devil_scene(res=256, film=64, device=dev)
grads = u.alloc_grads(scene)
for i in range(3):
    u.gradient_support(scene.medium.sigma_t, grads)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(10):
    s = u.gradient_support(scene.medium.sigma_t, grads)
torch.cuda.synchronize()
host = (time.perf_counter() - t) / 10 * 1e3
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(10):
    s = u.gradient_support(scene.medium.sigma_t, grads)
b.record(); torch.cuda.synchronize()
print("gradient_support: wall", round(host, 3), "ms per call, device", round(a.elapsed_time(b) / 10, 3), "ms; blocks", s.count, "of", s.mask.numel())
