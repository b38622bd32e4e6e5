"""LD_LIBRARY_PATH=variants/X python tools/gpu/share.py  -> step time of rank 0's share of the headline at factor 8 for G = 1, 4, 8 (one GPU)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import uivr_amd as u
from uivr_amd import synthetic
import bench
dev = torch.device('cuda', 0)
scene = synthetic.dust_devil_scene(res=256, film=512, device=dev)
scene.medium.majorant_resolution_factor = int(os.environ.get("FACTOR", "8"))
integ = u.get_int_config('volpathsimple-drt').create(max_depth=64)
res = []
for w in (1, 4, 8):
    sh = u.ShardSpec(0, w, u.ShardSpec.default_chunk(512 * 512, w)) if w > 1 else None
    r = bench.h1_rate(torch, u, scene, integ, 32, steps=10, warmup=3, shard=sh)
    res.append(f"G{w}: {r['ms_per_step']} ({r['t_primal_ms']}/{r['t_adjoint_ms']}/{r['t_grad_reduce_ms']})")
print(" | ".join(res))
