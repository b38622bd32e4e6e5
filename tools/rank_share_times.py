"""Step time of ONE rank's share of the headline (interleaved pixel chunks, ShardSpec(0, G)) for G = 1, 2, 4, 8, measured on
one GPU: the per-rank compute column of DESIGN.md section 7's projection (the all-reduce needs G GPUs).
    python tools/rank_share_times.py"""
import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uivr_amd as u
from uivr_amd import synthetic
import bench
dev = torch.device('cuda', 0)
scene = synthetic.dust_devil_scene(res=256, film=512, device=dev)
integ = u.get_int_config('volpathsimple-drt').create(max_depth=64)
for w in (1, 2, 4, 8):
    sh = u.ShardSpec(0, w, u.ShardSpec.default_chunk(512 * 512, w)) if w > 1 else None
    h = integ.native_handle(scene); h.enable_timing(True)
    r = bench.h1_rate(torch, u, scene, integ, 32, steps=10, warmup=3, shard=sh)
    tp, ta, tr = h.read_timings(0), h.read_timings(1), h.read_timings(2)
    print(w, r['ms_per_step'], 'primal', round(sum(tp)/len(tp),3), 'adj', round(sum(ta)/len(ta),3), 'red', round(sum(tr)/len(tr),3))
