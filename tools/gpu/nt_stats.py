"""LD_LIBRARY_PATH=variants/ntstats python tools/gpu/nt_stats.py -> window statistics of the nerf tile adjoint on config 5 (experiment build)"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import uivr_amd as u
from uivr_amd import synthetic
dev = torch.device('cuda', 0)
sc = synthetic.dust_devil_scene(res=256, film=512, device=dev)
sc.medium.emission = sc.medium.albedo
integ = u.get_int_config('nerf').create(max_depth=64)
spp = 32
img = u.render_primal(sc, integ, 0, spp, 7)
g = u.render_backward(sc, integ, ((2.0 / img.numel()) * (img - 0.5)).contiguous(), 0, spp, 7)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.environ.get("LD_LIBRARY_PATH", "").split(":")[0], "libdrt_hip.so"))
out = (ctypes.c_ulonglong * 8)()
print("rc", lib.drt_nt_debug_read(out, 0), "moves, direct splats, splats, workgroup steps, steps with misses:", list(out)[:5])
