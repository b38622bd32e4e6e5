#!/usr/bin/env python3
"""BASELINE config 3: the full optimisation loop (python/optimize.py:275-365) on the synthetic
dust-devil 256^3 density + albedo target, multi-sensor batched rays, DRT, Adam.  Reports iterations/s
and the loss trajectory.  Paper defaults (reproduce.py:45-59): batch 32768 px, spp_grad 16,
spp_primal 16*64, lr per scene; here n_iter is short - this measures loop throughput.

    python tools/bench_optimize.py [--res 256] [--iters 50] [--sensors 63]
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
    ap.add_argument("--sensors", type=int, default=63)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--primal-spp-factor", type=int, default=64)
    ap.add_argument("--ref-spp", type=int, default=256)
    ap.add_argument("--majorant-factor", type=int, default=8, help="reference default: 8 (scene_config.py:36); 0 = global majorant")
    args = ap.parse_args()
    import torch
    import uivr_amd as u
    from uivr_amd import synthetic
    dev = torch.device("cuda:0")
    scene = synthetic.dust_devil_scene(res=args.res, film=args.film, device=dev, n_sensors=args.sensors)
    scene.medium.majorant_resolution_factor = args.majorant_factor
    sc = u.SceneConfig(name="dust-devil-synthetic", scene=scene, param_keys=[u.SIGMA_T_KEY, u.ALBEDO_KEY],
                       sensors=list(range(args.sensors)), start_from_value={u.SIGMA_T_KEY: 0.04, u.ALBEDO_KEY: 0.6},
                       max_depth=64, ref_spp=args.ref_spp, majorant_resolution_factor=args.majorant_factor)
    oc = u.OptimizationConfig("config3", spp=args.spp, n_iter=args.iters, lr=3e-4 * 100, batch_size=args.batch,
                              primal_spp_factor=args.primal_spp_factor, lr_schedule=u.Schedule.Last25,
                              checkpoint_initial=False, checkpoint_final=False, checkpoint_stride=0)
    t_ref = time.perf_counter()
    integ = u.get_int_config(sc.ref_integrator).create(max_depth=64)
    refs = torch.stack([u.render_primal(scene, integ, s, args.ref_spp, 1234).view(args.film, args.film, 3)
                        for s in range(args.sensors)])
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t_ref
    stamps = []
    t0 = time.perf_counter()
    _, params, _, hist = u.run_optimization(None, oc, sc, "volpathsimple-drt", ref_images=refs,
                                            progress=lambda i, l: stamps.append(time.perf_counter()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steady = (stamps[-1] - stamps[4]) / (len(stamps) - 5) if len(stamps) > 10 else dt / args.iters
    rays = args.batch * args.spp * (args.primal_spp_factor + 2)       # primal + (primal replay + adjoint) at spp_grad
    print(json.dumps({"workload": f"optimize loop, dust-devil {args.res}^3, {args.sensors} sensors {args.film}^2, batch {args.batch}, "
                                  f"spp_grad {args.spp}, spp_primal {args.spp * args.primal_spp_factor}",
                      "it_per_s": round(1.0 / steady, 3), "ms_per_iteration": round(1e3 * steady, 2),
                      "Mrays_per_iteration": round(rays / 1e6, 2), "Mrays_per_s": round(rays / steady / 1e6, 1),
                      "reference_render_s": round(t_ref, 2), "loss_first5": [round(x, 5) for x in hist[:5]],
                      "loss_last5": [round(x, 5) for x in hist[-5:]]}))


if __name__ == "__main__":
    main()
