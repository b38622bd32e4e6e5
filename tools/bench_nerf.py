#!/usr/bin/env python3
"""BASELINE config 5: the `nerf` IntegratorConfig (emissive RGB + sigma_t grid, python/integrators/nerf.py)
on a 256^3 grid, 512^2 x spp, primal + adjoint (the H1 sequence).  4 gradient channels per query:
the atomic-add stress case.  Prints Msamples/s and the kernel times.

    python tools/bench_nerf.py [--res 256] [--film 512] [--spp 8] [--queries 64] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--film", type=int, default=512)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dense", action="store_true", help="fog everywhere (every query splats) instead of the sparse dust devil")
    ap.add_argument("--debug-flags", type=int, default=0)
    args = ap.parse_args()
    import torch
    import uivr_amd as u
    from uivr_amd import synthetic
    dev = torch.device("cuda:0")
    scene = synthetic.dust_devil_scene(res=args.res, film=args.film, device=dev)
    if args.dense:
        scene.medium.sigma_t.add_(0.05 * float(scene.medium.sigma_t.max()))
    scene.medium.emission = (scene.medium.albedo * 0.8 + 0.1).contiguous()
    integ = u.load_dict({"type": "nerf", "queries_per_ray": args.queries, "test_hooks": bool(args.debug_flags)})
    h = integ.native_handle(scene)
    if args.debug_flags:
        h.set_debug_flags(args.debug_flags)
    n = args.film * args.film
    spp = args.spp

    def step(i):
        seed = u.sample_tea_32(i, 77)[0]
        img = u.render_primal(scene, integ, 0, spp, seed)
        gi = ((2.0 / (n * 3)) * (img - 0.5)).contiguous()
        return u.render_backward(scene, integ, gi, 0, spp, seed)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    h.enable_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        g = step(args.warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    tp, ta, tr = h.read_timings(0), h.read_timings(1), h.read_timings(2)
    h.enable_timing(False)
    mean = lambda v: round(sum(v) / args.steps, 3)       # per step (sub-batch launches summed)
    print(json.dumps({"workload": f"nerf {args.res}^3 {'dense' if args.dense else 'dust-devil'}, {args.film}^2 x {spp} spp, {args.queries} queries/ray",
                      "Msamples_per_s": round(n * spp / dt / 1e6, 2), "ms_per_step": round(1e3 * dt, 3),
                      "t_primal_ms": mean(tp), "t_adjoint_ms": mean(ta), "t_grad_reduce_ms": mean(tr),
                      "grad_abs_sum": float(g[u.SIGMA_T_KEY].abs().sum())}))


if __name__ == "__main__":
    main()
