"""Step time of ONE rank's share of the headline (interleaved pixel chunks, ShardSpec(0, G)) for G = 1, 2, 4, 8, measured on
one GPU: the per-rank compute column of DESIGN.md section 7's projection (the all-reduce needs G GPUs).  At the reference's
majorant_resolution_factor 8 (queued supergrid tracer) and with the global majorant; `ordered`: the library flavour with
test hooks schedules launches of fewer than 1.5 M rays like large ones (ray order / tail launch: debug bit 1073741824).
    python tools/rank_share_times.py [out.json]"""
import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uivr_amd as u
from uivr_amd import synthetic
import bench
dev = torch.device('cuda', 0)
out = {}
for factor in (8, 0):
    scene = synthetic.dust_devil_scene(res=256, film=512, device=dev)
    scene.medium.majorant_resolution_factor = factor
    for ordered in (False, True):
        integ = u.get_int_config('volpathsimple-drt').create(max_depth=64)
        if ordered:                                         # the same properties, bound to the library flavour with test hooks
            integ = u.load_dict(dict(integ.props(), type="volpathsimple", test_hooks=True))
        h = integ.native_handle(scene)
        if ordered:
            h.set_debug_flags(1073741824)
        for w in (1, 2, 4, 8):
            sh = u.ShardSpec(0, w, u.ShardSpec.default_chunk(512 * 512, w)) if w > 1 else None
            h.enable_timing(True)
            r = bench.h1_rate(torch, u, scene, integ, 32, steps=10, warmup=3, shard=sh, roofline=False)
            key = f"factor{factor}{'_ordered' if ordered else ''}_G{w}"
            out[key] = {"ms_per_step": r['ms_per_step'], "rays_per_rank": 512 * 512 * 32 // w}
            print(key, out[key], flush=True)
        if ordered:
            h.set_debug_flags(0)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
