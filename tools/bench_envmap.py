"""H1 rate of the headline scene lit by an environment map (N4; all five scenes of the paper use one):
    python tools/bench_envmap.py [--debug-flags N] [--majorant-factor F]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import uivr_amd as u
from uivr_amd import synthetic
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--debug-flags", type=int, default=0)
ap.add_argument("--majorant-factor", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda", 0)
scene = synthetic.dust_devil_scene(res=256, film=512, device=dev)
scene.medium.majorant_resolution_factor = args.majorant_factor
g = torch.Generator().manual_seed(5)
pix = (torch.rand(256, 512, 3, generator=g) ** 4 * 3.0 + 0.2).to(dev)          # a few bright texels: the importance sampler matters
scene.emitter = u.EnvmapEmitter(pixels=pix, scale=1.0)
integ = u.get_int_config("volpathsimple-drt").create(max_depth=64, **({"test_hooks": True} if args.debug_flags else {}))
if args.debug_flags:
    integ.native_handle(scene).set_debug_flags(args.debug_flags)
h = integ.native_handle(scene)
h.enable_timing(True)
r = bench.h1_rate(torch, u, scene, integ, 32, steps=5, warmup=2)
r["t_primal_ms"] = sum(h.read_timings(0)) / 7; r["t_adjoint_ms"] = sum(h.read_timings(1)) / 7
print(json.dumps(r))
