#!/usr/bin/env python3
"""Repeat the same nerf / DRT backward pass and compare the gradients between repetitions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import uivr_amd as u
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "nerf"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
scene = u.cube_test_scene(32, 32, density_scale=1.5)
sg = u.scene_to(scene, dev)
if which == "nerf":
    integ = u.load_dict(dict(type="nerf", queries_per_ray=64, activation="relu", test_hooks=True))
else:
    integ = u.get_int_config("volpathsimple-drt").create(max_depth=64, test_hooks=True)
spp, seed = 4, 1234
integ.native_handle(sg).set_debug_flags(flags)
n = 32 * 32
img = u.render_primal(sg, integ, 0, spp, seed)
gi = ((2.0 / (n * 3)) * (img - 0.5)).contiguous()
ref = None
bad = 0
worst = {}
for r in range(reps):
    g = u.render_backward(sg, integ, gi, 0, spp, seed)
    cur = {k: v.detach().cpu().numpy().copy() for k, v in g.items() if not k.startswith("_")}
    if ref is None:
        ref = cur
        continue
    for k in cur:
        d = np.abs(cur[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)
        worst[k] = max(worst.get(k, 0.0), d)
        if d > 1e-5 and bad < 6:
            bad += 1
            idx = np.unravel_index(np.argmax(np.abs(cur[k] - ref[k])), cur[k].shape)
            print(f"rep {r} {k}: rel diff {d:.3e} at {idx}: {cur[k][idx]} vs {ref[k][idx]}; sums {cur[k].sum():.6e} {ref[k].sum():.6e}")
print("flags", flags, "worst rel diff vs rep 0:", worst)
