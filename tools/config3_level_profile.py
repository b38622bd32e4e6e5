#!/usr/bin/env python3
"""ONE resolution level of BASELINE config 3 as python/reproduce.py sets the dust-devil DRT run up (bench.py: config3_as_reproduce), for
kernel traces per level:

    rocprofv3 --kernel-trace --stats -d out -o lv -- python tools/config3_level_profile.py --res 16 --iters 60

Grid `res`^3 filled with the run's initial values (sigma_t 0.04 / 100, albedo 0.6: python/scene_config.py:166-169) - `--trained` instead
starts from the target volume resampled to that level, which is what the later iterations of a level look like -, Adam lr 3e-4, l1, batch
32768 px, spp 1024 / 16, majorant_resolution_factor 8 adjusted as optimize.py:182-199 does, 4096 x 2048 environment map, 63 sensors 512^2.
Prints iterations/s (device time) and the HOST time per iteration (how long the Python loop takes to enqueue one: the loop keeps no
device -> host wait per iteration, so the host may run ahead)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=16)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--factor", type=int, default=8)
    ap.add_argument("--trained", action="store_true")
    ap.add_argument("--no-envmap", action="store_true")
    a = ap.parse_args()
    import torch
    import uivr_amd as u
    from uivr_amd import synthetic
    dev = torch.device("cuda:0")
    target = synthetic.dust_devil_scene(res=256, film=512, device=dev, n_sensors=63)
    target.medium.majorant_resolution_factor = a.factor
    if not a.no_envmap:
        g = torch.Generator().manual_seed(5)
        target.emitter = u.EnvmapEmitter(pixels=(torch.rand(2048, 4096, 3, generator=g) ** 4 * 3.0 + 0.2).to(dev), scale=1.0)
    scfg = u.SceneConfig(name="dust-devil", scene=target, param_keys=[u.SIGMA_T_KEY, u.ALBEDO_KEY], sensors=list(range(63)),
                         start_from_value={u.SIGMA_T_KEY: 0.04 / 100, u.ALBEDO_KEY: 0.6}, majorant_resolution_factor=a.factor, ref_spp=16)
    rendered = u.render_reference_image(scfg, {s_: None for s_ in scfg.sensors})
    ref = torch.stack([rendered[s_] for s_ in scfg.sensors])
    # the level's grid: the target's lattice coarsened to res^3 (run_optimization initialises on the scene's lattice / 2^len(upsample))
    from uivr_amd.optimize import upsample_grid
    lv = synthetic.dust_devil_scene(res=a.res, film=512, device=dev, n_sensors=63)
    lv.emitter = target.emitter
    lv.medium.majorant_resolution_factor = a.factor
    if a.trained:
        start = {u.SIGMA_T_KEY: None, u.ALBEDO_KEY: None}
    else:
        start = {u.SIGMA_T_KEY: 0.04 / 100, u.ALBEDO_KEY: 0.6}
    scfg_lv = u.SceneConfig(name="dust-devil", scene=lv, param_keys=[u.SIGMA_T_KEY, u.ALBEDO_KEY], sensors=list(range(63)),
                            start_from_value=start, majorant_resolution_factor=a.factor, ref_spp=16)
    out = {}
    # (every run_optimization call builds its integrator - scratch allocations, the first launch of every kernel: the first SKIP iterations
    #  are left out of the rate)
    SKIP = 6
    for n_iter in (8, a.iters + SKIP):
        oc = u.OptimizationConfig(name="lv", spp=16, n_iter=n_iter, lr=3e-4, primal_spp_factor=64, batch_size=32768)
        marks = {}

        def prog(i, l):
            if i == SKIP - 1:
                torch.cuda.synchronize()
                marks["t0"] = time.perf_counter()
            if i == n_iter - 1:
                marks["t_host_end"] = time.perf_counter()     # (the loop has enqueued its last iteration; run_optimization's loss read-back follows)

        torch.cuda.synchronize()
        u.run_optimization(None, oc, scfg_lv, "volpathsimple-drt", ref_images=ref, progress=prog)
        t_host = marks.get("t_host_end", time.perf_counter()) - marks.get("t0", time.perf_counter())
        torch.cuda.synchronize()
        dt = time.perf_counter() - marks.get("t0", time.perf_counter())
        n = n_iter - SKIP
        if n > 0:
            out = {"grid": f"{a.res}^3", "trained": a.trained, "iterations": n, "iterations_per_s": round(n / dt, 2),
                   "ms_per_iteration": round(1e3 * dt / n, 3), "host_ms_per_iteration": round(1e3 * t_host / n, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
