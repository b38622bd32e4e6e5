#!/usr/bin/env python3
"""bench.py -- Msamples/s primal+adjoint DRT on the BASELINE headline workload.

One *step* = one H1 pass (reference: python/batched.py:255-326 with spp_grad = spp)
over the whole image: primal tracing kernel -> box film -> loss gradient
(mean((img-0.5)^2), tests/test_integrators.py:119) -> dL -> adjoint tracing kernel
(consumes the stored primal radiance as state_in) [-> one all-reduce of the
gradient grids when N > 1].  One *sample* = one camera ray (pixel x spp).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the keys below).  Inputs (the synthetic 256^3
dust-devil grids) are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); 6290 measured copy


def kernel_source_sha16():
    """sha256[:16] over the kernel / C-ABI sources (csrc/*.hip, *.h, *.cpp, include/*.h): ties a committed counter profile
    to the code it was taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "unbiased-inverse-volume-rendering_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "unbiased-inverse-volume-rendering_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "unbiased-inverse-volume-rendering_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_traffic(key):
    """(bytes per adjoint pass, source) of the committed counter profile for workload `key`, or (None, None) when the
    profile was taken from other kernel sources than the ones this run executes."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("source_sha16") == kernel_source_sha16() and tj.get(key) is not None:
                return tj.get(key), (f"profiles/roofline_traffic.json (rocprofv3 PMC passes over these kernel sources, "
                                     f"sha16 {tj.get('source_sha16')})")
        except Exception:
            pass
    return None, None


LDS_ADD_U64_PEAK = 3.4e12          # ds_add_u64 on random addresses of a 9.5 k-entry tile, whole chip (tools/ubench/lds_atomic_rate.hip)


def committed_secondary(key, counters_adjoint):
    """SURVEY.md 8d's secondary ceilings for workload `key`, from the committed utilisation counter pass
    (tools/pmc_to_util.py; hash-gated like `committed_traffic`): lanes active per vector instruction and VALU-busy of the
    tracer kernels, and the LDS-atomic rate of tile_reduce (its adds follow from this run's event counters: 8 corners per
    sigma_t record, 24 more per colour record) against the measured ds_add_u64 ceiling.  None without a matching profile."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        with open(tpath) as f:
            tj = json.load(f)
    except Exception:
        return None
    util = tj.get("_util:" + key)
    if tj.get("source_sha16") != kernel_source_sha16() or not util:
        return None
    out = {"source": f"profiles/roofline_traffic.json _util:{key} (rocprofv3 PMC pass + kernel trace over these kernel sources)",
           "kernels": {}}
    for k, v in util.items():
        counting = "<true, true" in k or "<false, true" in k                      # (the instantiation the counter step runs)
        if ("trace_" in k or "nerf_tile_adjoint" in k or "nerf_kernel" in k) and not counting and (v.get("avg_ms") or 0) > 0.05:
            out["kernels"][k] = {"avg_ms": v["avg_ms"], "lanes_active": v["lanes_active"], "valu_busy": v["valu_busy"]}
    tr = next((v for k, v in util.items() if "tile_reduce_kernel" in k), None)
    if tr and (tr.get("ms_per_step") or tr.get("avg_ms")) and counters_adjoint:
        c = counters_adjoint
        adds = 8 * (c["n_tr"] + c["n_rt_adj"] + c["n_sc"]) + 24 * c["n_sc_alb"]
        t_ms = tr.get("ms_per_step") or tr["avg_ms"]               # (a step may run several reduce launches: ray sub-batches)
        rate = adds / (t_ms * 1e-3)
        out["tile_reduce"] = {"ms_per_step": t_ms, "lds_adds_per_step": adds, "lds_add_rate": round(rate / 1e12, 3),
                              "unit": "T adds/s", "peak": LDS_ADD_U64_PEAK / 1e12, "frac": round(rate / LDS_ADD_U64_PEAK, 4),
                              "lanes_active": tr["lanes_active"], "valu_busy": tr["valu_busy"]}
    return out


def algorithmic_bytes(cnt, n_samples, primal_io=True, adjoint_io=True):
    """32 B per sigma_t lookup, 96 B per albedo lookup, 64 B per sigma_t
    splat, 192 B per albedo splat; ray I/O: primal out 12 B L (+24 B o,d when rays are
    read); adjoint in 12 B dL + 12 B state_in."""
    b = 32 * (cnt["n_dt"] + cnt["n_rt"] + cnt["n_drt"]) + 96 * cnt["n_alb"] \
        + 64 * (cnt["n_tr"] + cnt["n_rt_adj"] + cnt["n_sc"]) + 192 * cnt["n_sc_alb"]
    io = 0
    if primal_io:
        io += 12 + 24
    if adjoint_io:
        io += 24
    return b + io * n_samples


def h1_rate(torch, u, scene, integ, spp, steps=5, warmup=2, shard=None, keys=None, roofline=True, traffic_key=None):
    """Msamples/s of H1 steps (primal -> film -> loss gradient -> adjoint) over the local rays of `shard`, and - with
    `roofline` - the same evidence as the headline line carries: HIP-event times of the tracing launches and of the
    gradient reduction, the event counters of one step, the algorithmic bytes they stand for and the
    achieved fraction of the HBM roofline of the adjoint pass (tracer + reductions) and of the primal launch."""
    sensor = scene.sensors[0]
    n_pixels = sensor.width * sensor.height
    sh = shard or u.ShardSpec()
    n_local = sh.n_local_pixels(n_pixels)
    off, inter = sh.ray_mapping(spp)
    batch = u.RayBatch(n_rays=n_local * spp, spp=spp, sensor=sensor, ray_offset=off, interleave=inter)
    grads = u.alloc_grads(scene, keys or integ.param_keys)
    loss_scale = 2.0 / (n_pixels * 3)
    h = integ.native_handle(scene)

    def step(i):
        sampler = u.IndependentSampler(u.sample_tea_32(2 * i + 1, 988378)[0], spp)
        grads["_flat"].zero_()
        L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)
        img = integ.develop(scene, L, spp)
        dL = integ.film_backward(scene, loss_scale * (img - 0.5), spp)
        integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)
        return sampler, state, dL

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    h.enable_timing(True)
    # two timed blocks of `steps` steps; `value` is the MEAN over both (as rounds 1-3 reported it; round 4's entries quoted the faster block,
    # which biased every side entry upward - advisor r4); the faster block stays as `best_block_msamples_per_s` (a side entry's few steps
    # are short enough for one host hiccup - seen once after the CPU-baseline leg: 14.4 ms of wall clock per step around 6.5 ms of
    # kernels - to halve a block's rate).  The HIP-event kernel times below are sums over BOTH blocks divided by their launch counts.
    blocks = []
    for b in range(2):
        t0 = time.perf_counter()
        for i in range(steps):
            sampler, state, dL = step(warmup + b * steps + i)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / steps)
    dt = sum(blocks) / len(blocks)
    t_p, t_a, t_r, t_pass = (h.read_timings(k) for k in range(4))
    h.enable_timing(False)
    out = {"value": round(n_local * spp / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3),
           "n_samples_per_step": n_local * spp, "blocks_msamples_per_s": [round(n_local * spp / b / 1e6, 2) for b in blocks],
           "best_block_msamples_per_s": round(n_local * spp / min(blocks) / 1e6, 2)}
    if roofline:
        n = n_local * spp
        h.enable_counters(True)
        h.reset_counters()
        integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)
        cp = {k: int(v) for k, v in h.get_counters().items()}
        h.reset_counters()
        grads["_flat"].zero_()
        integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)
        ca = {k: int(v) for k, v in h.get_counters().items()}
        h.enable_counters(False)
        b_p = algorithmic_bytes(cp, n, primal_io=True, adjoint_io=False) - 24 * n        # rays generated on device
        b_a = algorithmic_bytes(ca, n, primal_io=False, adjoint_io=True)
        per = max(1, len(t_pass))
        avg_p = sum(t_p) / max(1, len(t_p))
        avg_pass, avg_a, avg_r = sum(t_pass) / per, sum(t_a) / per, sum(t_r) / per
        ach_a = b_a / (avg_pass * 1e-3) / 1e9 if avg_pass > 0 else 0.0
        ach_p = b_p / (avg_p * 1e-3) / 1e9 if avg_p > 0 else 0.0
        out.update({"t_primal_ms": round(avg_p, 3), "t_adjoint_ms": round(avg_a, 3), "t_grad_reduce_ms": round(avg_r, 3),
                    "counters_primal": cp, "counters_adjoint": ca,
                    "roofline": {"bound": "hbm", "kernel": "adjoint pass (tracer + record partition + tile_reduce)",
                                 "achieved": round(ach_a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_a / HBM_PEAK_GBS, 5),
                                 "traffic": committed_traffic(traffic_key)[0] if traffic_key else None,
                                 "secondary": committed_secondary(traffic_key, ca) if traffic_key else None,
                                 "algorithmic_bytes_per_launch": b_a, "avg_launch_ms": round(avg_pass, 4),
                                 "sum_tracer_ms": round(avg_a, 4), "sum_reduction_ms": round(avg_r, 4),
                                 "bytes_per_sample_h1": round((b_p + b_a) / n, 1),
                                 "primal": {"achieved": round(ach_p, 2), "frac": round(ach_p / HBM_PEAK_GBS, 5),
                                            "algorithmic_bytes_per_launch": b_p, "avg_launch_ms": round(avg_p, 4)}}})
    h.release_scratch()
    return out


def other_configs(torch, u, synthetic, dev, integ_name="volpathsimple-drt", only=None, main_factor=8):
    """BASELINE.json `configs` 2-5 and the reference's default majorant_resolution_factor on the headline scene,
    each at its registered size, a few H1 steps each (wall clock around synchronised steps, 1 GPU)."""
    out = {}

    def guarded(name, fn):
        if only and name != only:
            return
        try:
            out[name] = fn()
        except Exception as e:                     # a failure here must not take the headline line with it
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    def envmap_pixels(w, h):
        g = torch.Generator().manual_seed(5)
        return (torch.rand(h, w, 3, generator=g) ** 4 * 3.0 + 0.2).to(dev)          # a few bright texels: the importance sampler matters

    def cfg2():
        # the reference's scenes all run majorant_resolution_factor 8 (scene_config.py:36): that is the entry's value; the
        # global majorant beside it
        sc = synthetic.smoke_scene(res=128, film=512, device=dev)
        sc.medium.majorant_resolution_factor = 8
        r = h1_rate(torch, u, sc, u.get_int_config(integ_name).create(max_depth=64), 16)
        sc.medium.majorant_resolution_factor = 0
        r["global_majorant"] = h1_rate(torch, u, sc, u.get_int_config(integ_name).create(max_depth=64), 16, roofline=False)
        r["workload"] = "config 2: smoke plume 128^3 (janga-smoke stand-in), 512x512x16spp, majorant_resolution_factor 8 (16^3 supergrid)"
        return r

    def other_factor():
        # the headline scene at the majorant setting the main line does NOT run (main line: the reference's default 8)
        f = 0 if main_factor else 8
        sc = synthetic.dust_devil_scene(res=256, film=512, device=dev)
        sc.medium.majorant_resolution_factor = f
        r = h1_rate(torch, u, sc, u.get_int_config(integ_name).create(max_depth=64), 32, steps=10, warmup=3,
                    traffic_key="dust-devil-256-512x32" + ("-factor8" if f else ""))
        r["workload"] = ("headline scene at majorant_resolution_factor 8 (the reference's default, scene_config.py:36): queued supergrid tracer drt_sq.hip"
                         if f else "headline scene with ONE global majorant (majorant_resolution_factor 0): wave-cooperative tracer drt_coop.hip")
        return r

    def envmap8():
        # what the paper's scenes run (scene_config.py:36,102,152): majorant supergrid AND an environment map of that size
        sc = synthetic.dust_devil_scene(res=256, film=512, device=dev)
        sc.medium.majorant_resolution_factor = 8
        sc.emitter = u.EnvmapEmitter(pixels=envmap_pixels(2048, 1024), scale=1.0)
        r = h1_rate(torch, u, sc, u.get_int_config(integ_name).create(max_depth=64), 32, steps=10, warmup=3,
                    traffic_key="dust-devil-256-512x32-factor8-envmap2048")
        r["workload"] = ("headline scene as the reference's scenes are set up: majorant_resolution_factor 8 and a 2048x1024 environment map "
                         "(importance-sampled lat-long map, MIS with phase sampling)")
        return r

    def envmap():
        sc = synthetic.dust_devil_scene(res=256, film=512, device=dev)
        sc.emitter = u.EnvmapEmitter(pixels=envmap_pixels(512, 256), scale=1.0)
        r = h1_rate(torch, u, sc, u.get_int_config(integ_name).create(max_depth=64), 32)
        r["workload"] = "headline scene lit by a 512x256 environment map, global majorant (N4: importance-sampled lat-long map, MIS with phase sampling)"
        return r

    def cfg4():
        # the reference's default majorant_resolution_factor 8 on a 512^3 grid: a 64^3 supergrid - its majorants do not fit LDS
        # next to the ray records, the queued tracer's MG kernels keep one bit per cell there and read the majorants from L2
        sc = synthetic.dust_devil_scene(res=512, film=1024, device=dev)
        sc.medium.majorant_resolution_factor = 8
        r = h1_rate(torch, u, sc, u.get_int_config(integ_name).create(max_depth=64), 64, steps=3, warmup=1,
                    shard=u.ShardSpec(0, 8, 2048), traffic_key="config4-512-1024x64-rank0of8-factor8")
        sc.medium.majorant_resolution_factor = 0
        r["global_majorant"] = h1_rate(torch, u, sc, u.get_int_config(integ_name).create(max_depth=64), 64, steps=3, warmup=1,
                                       shard=u.ShardSpec(0, 8, 2048), roofline=False)
        # G = 8 projection: the rank's MEASURED compute + the one-collective all-reduce priced from xGMI link rates (ring over one
        # 153 GB/s link <-> direct over 7 links) at the size gradient_support leaves of the 2 GiB buffer, + pack / unpack as measured at this size
        try:
            g4 = u.alloc_grads(sc)
            sup = u.gradient_support(sc.medium.sigma_t, g4)
            frac = sup.count / sup.mask.numel() if sup is not None else 1.0
            nbytes = g4["_flat"].numel() * 4 * (frac if frac <= 0.7 else 1.0)
            del g4, sup
            ring, direct = 2 * 7 / 8 * nbytes / 153e9 * 1e3, 2 * 7 / 8 * nbytes / (7 * 153e9) * 1e3
            r["projection_G8_UNMEASURED_ON_MULTI_GPU"] = {
                "rank_compute_ms": r["ms_per_step"], "allreduce_fraction_of_2GiB": round(frac if frac <= 0.7 else 1.0, 3),
                "allreduce_ms_ring_vs_direct": [round(ring, 2), round(direct, 2)],
                "step_ms_ring_vs_direct": [round(r["ms_per_step"] + ring, 2), round(r["ms_per_step"] + direct, 2)],
                "msamples_per_s_8gpu_ring_vs_direct": [round(1024 * 1024 * 64 / (r["ms_per_step"] + ring) / 1e3, 1),
                                                       round(1024 * 1024 * 64 / (r["ms_per_step"] + direct) / 1e3, 1)]}
        except Exception as e:                          # (a projection must not take the entry with it)
            r["projection_G8_UNMEASURED_ON_MULTI_GPU"] = {"error": f"{type(e).__name__}: {e}"}
        r["workload"] = ("config 4: 512^3 grid, rank 0's share (1/8, interleaved 2048-pixel chunks) of 1024x1024x64spp; "
                         "per-GPU compute only, the 2 GiB gradient all-reduce is not included; majorant_resolution_factor 8 "
                         "(64^3 supergrid: drt_sq.hip, majorants from L2); global_majorant: the same with ONE majorant")
        return r

    def cfg5():
        sc = synthetic.dust_devil_scene(res=256, film=512, device=dev)
        sc.medium.emission = sc.medium.albedo
        integ = u.get_int_config("nerf").create(max_depth=64)
        r = h1_rate(torch, u, sc, integ, 32, steps=3, warmup=1, roofline=False)
        r["workload"] = "config 5 (nerf IntegratorConfig, 128 queries): 256^3 sigma_t + emission grids, 512x512x32spp"
        # The adjoint of sensor rays (drt_nerf_tile.hip) pre-reduces its splats in an LDS window: its ceiling is the LDS atomic rate, not HBM (the
        # SURVEY 8d byte model prices every march step as an independent 8-corner access and comes out above 1).  Its roofline: LDS lane-adds of one
        # step (8 per non-zero plane of a query's splat, counted by the kernel) over the launch's HIP-event time, against the measured ds_add_u64
        # rate (tools/ubench/lds_atomic_conflict_rate.hip: 3.2 T lane-adds/s at 64 distinct addresses per instruction, 1.4 / 2.0 / 2.9 at 8 / 16 / 32).
        sensor = sc.sensors[0]
        n = sensor.width * sensor.height * 32
        batch = u.RayBatch(n_rays=n, spp=32, sensor=sensor)
        grads = u.alloc_grads(sc, integ.param_keys)
        h = integ.native_handle(sc)
        sampler = u.IndependentSampler(u.sample_tea_32(77, 988378)[0], 32)
        L, _, state = integ.sample(u.ADMode.Primal, sc, sampler.clone(), batch)
        dL = integ.film_backward(sc, (2.0 / (n // 32 * 3)) * (integ.develop(sc, L, 32) - 0.5), 32)
        integ.sample(u.ADMode.Backward, sc, sampler.clone(), batch, δL=dL, state_in=state, grads=grads)      # warm
        torch.cuda.synchronize()
        h.enable_timing(True)
        for _ in range(3):
            integ.sample(u.ADMode.Backward, sc, sampler.clone(), batch, δL=dL, state_in=state, grads=grads)
        t_a = h.read_timings(1)
        h.enable_timing(False)
        h.enable_counters(True)
        h.reset_counters()
        integ.sample(u.ADMode.Backward, sc, sampler.clone(), batch, δL=dL, state_in=state, grads=grads)
        ca = {k: int(v) for k, v in h.get_counters().items()}
        adds = int(h.nerf_tile_lds_adds())
        h.enable_counters(False)
        ms = sum(t_a) / max(1, len(t_a))
        rate = adds / (ms * 1e-3) if ms > 0 else 0.0
        sec = committed_secondary("fused-256-512x32", None)
        tile_util = next((v for k, v in ((sec or {}).get("kernels") or {}).items() if "nerf_tile_adjoint" in k), None)
        r["roofline_adjoint"] = {"bound": "lds_atomics", "kernel": "nerf_tile_adjoint_kernel (+ its bounds reduction)", "queries_per_step": ca["n_dt"],
                                 "lds_lane_adds_per_step": adds, "avg_launch_ms": round(ms, 3), "achieved": round(rate / 1e12, 3), "unit": "T lane-adds/s",
                                 "peak": 3.2, "frac": round(rate / 3.2e12, 4),
                                 "peak_by_distinct_addresses_per_instruction": {"8": 1.39, "16": 2.05, "32": 2.89, "64": 3.24},
                                 # vector-instruction issue of the same kernel (committed counter pass of the fused configuration, which runs this kernel)
                                 "valu": tile_util}
        h.release_scratch()
        return r

    def cfg3():
        # loop throughput (it/s) AND the loss decrease over 200 iterations from the constant init, against
        # reference renderings of the target volume (optimize.py:56-87, 325-358; reproduce.py:45-59), at the reference's
        # default majorant_resolution_factor 8 and, for comparison, with the global majorant
        target = synthetic.dust_devil_scene(res=256, film=512, device=dev, n_sensors=63)
        const_emitter = target.emitter
        res = {}
        # (factor, environment map): the reference's set-up is (8, a 2k environment map), scene_config.py:36,102
        for factor, env in ((8, False), (0, False), (8, True)):
            target.medium.majorant_resolution_factor = factor
            target.emitter = u.EnvmapEmitter(pixels=envmap_pixels(2048, 1024), scale=1.0) if env else const_emitter
            scfg = u.SceneConfig(name="dust-devil", scene=target, param_keys=[u.SIGMA_T_KEY, u.ALBEDO_KEY], sensors=list(range(63)),
                                 start_from_value={u.SIGMA_T_KEY: 0.04, u.ALBEDO_KEY: 0.6}, majorant_resolution_factor=factor,
                                 ref_spp=64)
            rkey = "ref_env" if env else "ref"
            if rkey not in res:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rendered = u.render_reference_image(scfg, {s_: None for s_ in scfg.sensors})
                ref = torch.stack([rendered[s_] for s_ in scfg.sensors])
                torch.cuda.synchronize()
                res[rkey] = ref
                if not env:
                    res["reference_render_s"] = round(time.perf_counter() - t0, 2)
            r = {}
            for n_iter in (10, 200):                   # warm-up run, then the timed one
                oc = u.OptimizationConfig(name="b", spp=16, n_iter=n_iter, lr=5e-3, primal_spp_factor=64, batch_size=32768)
                stamps = []
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, _, _, hist = u.run_optimization(None, oc, scfg, integ_name, ref_images=res[rkey],
                                                   progress=lambda i, l: stamps.append(time.perf_counter()))
                torch.cuda.synchronize()
                r = {"value": round(n_iter / (time.perf_counter() - t0), 2), "unit": "iterations/s", "n_iter": n_iter,
                     "loss_first20_mean": round(sum(hist[:20]) / max(1, len(hist[:20])), 6),
                     "loss_last20_mean": round(sum(hist[-20:]) / max(1, len(hist[-20:])), 6)}
            r["loss_decreased"] = bool(r["loss_last20_mean"] < r["loss_first20_mean"])
            res[f"factor{factor}" + ("_envmap" if env else "")] = r
        target.emitter = const_emitter
        ref = res.pop("ref_env"); del ref
        ref = res.pop("ref"); del ref
        out3 = dict(res["factor8"])
        out3["global_majorant"] = res["factor0"]
        out3["envmap_factor8"] = res["factor8_envmap"]      # lit by a 2048x1024 environment map: the reference's scene set-up
        out3["reference_render_s"] = res["reference_render_s"]
        out3["workload"] = ("config 3: full optimisation loop, dust devil 256^3, 63 sensors 512^2, batch 32768 px, spp_grad 16, "
                            "spp_primal 1024, Adam lr 5e-3, l1, constant init (0.04, 0.6), 200 iterations against reference renderings "
                            "of the target volume (64 spp), majorant_resolution_factor 8 = the reference's default (reproduce.py:45-59)")
        out3["msamples_per_s"] = round(out3["value"] * 32768 * (1024 + 2 * 16) / 1e6, 1)
        return out3

    def rank_share():
        # what ONE rank of a G-GPU run of the headline computes per step (rank 0's interleaved pixel chunks, majorant_resolution_factor 8), timed
        # on this one GPU: the per-rank compute column of DESIGN.md section 7's projection (the all-reduce needs G GPUs).  A small launch is
        # bound by the latency of its longest paths, not by its work: G = 8 is an eighth of the rays and a third of the time.
        sc = synthetic.dust_devil_scene(res=256, film=512, device=dev)
        sc.medium.majorant_resolution_factor = 8
        integ = u.get_int_config(integ_name).create(max_depth=64)
        res = {}
        for w in (1, 2, 4, 8):
            sh = u.ShardSpec(0, w, u.ShardSpec.default_chunk(512 * 512, w)) if w > 1 else None
            r = h1_rate(torch, u, sc, integ, 32, steps=8, warmup=3, shard=sh, roofline=False)
            res[f"G{w}"] = {"ms_per_step": r["ms_per_step"], "rays_per_rank": r["n_samples_per_step"],
                            "msamples_per_s_per_rank": r["value"], "best_block_ms_per_step": round(r["n_samples_per_step"] / r["best_block_msamples_per_s"] / 1e3, 3)}
        res["G8_over_ideal"] = round(res["G8"]["ms_per_step"] / (res["G1"]["ms_per_step"] / 8), 2)
        res["workload"] = ("headline (dust devil 256^3, 512x512x32spp, factor 8): rank 0's share of G = 1 / 2 / 4 / 8 ranks, per-GPU compute only "
                           "(no all-reduce); UNMEASURED ON MULTI-GPU HARDWARE")
        res["value"], res["unit"] = res["G8"]["ms_per_step"], "ms per step of a rank's share at G = 8"
        return res

    def cfg3_reproduce():
        # The dust-devil DRT run as python/reproduce.py sets it up (:48-59, :108-110): Adam lr 3e-4 with the Last25 schedule, l1, batch 32768 px,
        # spp_grad 16, spp_primal 1024, constant init sigma_t 0.04 / 100, albedo 0.6 (scene_config.py:166-169) on a grid 2^4 times coarser than
        # the target, upsampled x2 at 4 / 16 / 36 / 64 % of the run (optimize.py:134-166, 228-252), majorant_resolution_factor 8 reduced on the
        # coarse grids as adjust_majorant_res_factor does (optimize.py:182-199), lit by a 4096 x 2048 environment map (scene_config.py:152).
        # The reference runs 6000 iterations; the bench runs 250 with the same fractions: iterations/s per resolution level.
        target = synthetic.dust_devil_scene(res=256, film=512, device=dev, n_sensors=63)
        target.medium.majorant_resolution_factor = 8
        target.emitter = u.EnvmapEmitter(pixels=envmap_pixels(4096, 2048), scale=1.0)
        scfg = u.SceneConfig(name="dust-devil", scene=target, param_keys=[u.SIGMA_T_KEY, u.ALBEDO_KEY], sensors=list(range(63)),
                             start_from_value={u.SIGMA_T_KEY: 0.04 / 100, u.ALBEDO_KEY: 0.6}, majorant_resolution_factor=8, ref_spp=64)
        rendered = u.render_reference_image(scfg, {s_: None for s_ in scfg.sensors})
        ref = torch.stack([rendered[s_] for s_ in scfg.sensors])
        out = {}
        for n_iter in (25, 250):                                   # warm-up run (every level once), then the timed one
            oc = u.OptimizationConfig(name="r", spp=16, n_iter=n_iter, lr=3e-4, primal_spp_factor=64, batch_size=32768,
                                      lr_schedule=u.Schedule.Last25, upsample=[0.04, 0.16, 0.36, 0.64])
            stamps = []
            # (the loop keeps no device -> host wait per iteration: the stamps that delimit a level wait for the device themselves)
            sync_at = {1, n_iter} | {int(x) for x in oc.upsample_at}

            def prog(i, l):
                if (i + 1) in sync_at:
                    torch.cuda.synchronize()
                stamps.append(time.perf_counter())

            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, params, _, hist = u.run_optimization(None, oc, scfg, integ_name, ref_images=ref, progress=prog)
            torch.cuda.synchronize()
            total = time.perf_counter() - t0
            bounds = [0] + sorted(oc.upsample_at) + [n_iter]
            levels = []
            for lv in range(len(bounds) - 1):
                a, b = bounds[lv], bounds[lv + 1]
                # (the first iteration of a level pays the upsampling and the scene rebuild: counted with its level)
                # (... except the run's very first iteration, which pays allocations and the first launches of every kernel)
                t_a = stamps[0] if a == 0 else stamps[a - 1]
                n_lv = b - a - (1 if a == 0 else 0)
                res = 256 >> (len(bounds) - 2 - lv)
                t_lv = stamps[b - 1] - t_a                           # (the warm-up run's first level is that one iteration: no rate)
                levels.append({"grid": f"{res}^3", "iterations": b - a, "iterations_per_s": round(n_lv / t_lv, 2) if n_lv > 0 and t_lv > 0 else None})
            out = {"value": round(n_iter / total, 2), "unit": "iterations/s", "n_iter": n_iter, "levels": levels,
                   "loss_first20_mean": round(sum(hist[:20]) / 20, 6), "loss_last20_mean": round(sum(hist[-20:]) / 20, 6),
                   "final_grid": list(params[u.SIGMA_T_KEY].shape)}
        out["slowest_level"] = min((l for l in out["levels"] if l["iterations_per_s"]), key=lambda l: l["iterations_per_s"])["grid"]
        out["workload"] = ("config 3 as python/reproduce.py runs dust-devil DRT: Adam lr 3e-4 (Last25), l1, batch 32768 px, spp 1024 / 16, init "
                           "sigma_t 0.04/100 and albedo 0.6 on 16^3, x2 upsampling at 4/16/36/64 % of the run, majorant_resolution_factor 8 "
                           "(adjusted on the coarse grids), 4096x2048 environment map, 63 sensors 512^2; 250 iterations instead of 6000")
        return out

    def cfg5_fused(env=False, factor=0):
        sc = synthetic.dust_devil_scene(res=256, film=512, device=dev)
        sc.medium.majorant_resolution_factor = factor
        if env:
            sc.emitter = u.EnvmapEmitter(pixels=envmap_pixels(2048, 1024), scale=1.0)
        integ = u.get_int_config("nerf-drt-fused").create(max_depth=64)
        sensor = sc.sensors[0]
        n_pixels, spp = sensor.width * sensor.height, 32
        batch = u.RayBatch(n_rays=n_pixels * spp, spp=spp, sensor=sensor)
        grads = u.alloc_grads(sc, integ.param_keys)
        loss_scale = 2.0 / (n_pixels * 6)
        h = integ.native_handle(sc)

        def step(i):
            sampler = u.IndependentSampler(u.sample_tea_32(2 * i + 1, 988378)[0], spp)
            grads["_flat"].zero_()
            L, _, state = integ.sample(u.ADMode.Primal, sc, sampler.clone(), batch)
            img = integ.develop(sc, L, spp)
            dL = integ.film_backward(sc, loss_scale * (img - 0.5), spp)
            integ.sample(u.ADMode.Backward, sc, sampler, batch, δL=dL, state_in=state, grads=grads)
            return sampler, state, dL

        step(0)
        torch.cuda.synchronize()
        h.enable_timing(True)
        t0 = time.perf_counter()
        steps = 3
        for i in range(steps):
            sampler, state, dL = step(1 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        t_p, t_pass = h.read_timings(0), h.read_timings(3)
        h.enable_timing(False)
        if env or factor:                            # the side entry: rate and pass times only
            h.release_scratch()
            return {"value": round(batch.n_rays / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3),
                    "t_primal_ms": round(sum(t_p) / steps, 3) if t_p else None,
                    "t_adjoint_pass_ms": round(sum(t_pass) / steps, 3) if t_pass else None}
        # event counts of one step -> algorithmic bytes with FOUR-CHANNEL events for the nerf queries (SURVEY.md 8d:
        # 128 B per four-channel lookup, 256 B per four-channel splat); the volpathsimple events as for the headline.
        # n_q = the nerf half's queries = fused sigma_t lookups - those of a volpathsimple-only pass over the same rays.
        h.enable_counters(True)
        h.reset_counters()
        integ.sample(u.ADMode.Primal, sc, sampler.clone(), batch)
        cp = {k: int(v) for k, v in h.get_counters().items()}
        h.reset_counters()
        grads["_flat"].zero_()
        integ.sample(u.ADMode.Backward, sc, sampler, batch, δL=dL, state_in=state, grads=grads)
        ca = {k: int(v) for k, v in h.get_counters().items()}
        h.enable_counters(False)
        drt = u.get_int_config(integ_name).create(max_depth=64)
        hd = drt.native_handle(sc)
        hd.enable_counters(True)
        hd.reset_counters()
        drt.sample(u.ADMode.Primal, sc, sampler.clone(), batch)
        n_q = cp["n_dt"] - int(hd.get_counters()["n_dt"])
        hd.enable_counters(False)
        n = batch.n_rays

        def bytes_of(c, adj):
            d = dict(c)
            d["n_dt"] -= n_q; d["n_alb"] -= n_q
            if adj:
                d["n_sc"] -= n_q; d["n_sc_alb"] -= n_q
            b = algorithmic_bytes(d, n, primal_io=not adj, adjoint_io=adj) - (24 * n if not adj else 0)   # rays generated on device
            return b + 128 * n_q + (256 * n_q if adj else 0) + (12 * n if not adj else 24 * n)          # + the nerf half's L / (dL, L_in)

        b_p, b_a = bytes_of(cp, False), bytes_of(ca, True)
        avg_pass = sum(t_pass) / steps if t_pass else 0.0
        avg_p = sum(t_p) / steps if t_p else 0.0
        h.release_scratch(); hd.release_scratch()
        return {"value": round(n / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3), "n_samples_per_step": n,
                "nerf_queries_per_step": n_q, "t_primal_ms": round(avg_p, 3), "t_adjoint_pass_ms": round(avg_pass, 3),
                "roofline_adjoint": {"bound": "hbm",
                                     # `frac`: the HBM-side COUNTER traffic over the pass time (what the memory system moved);
                                     # the SURVEY 8d lookup-rate figure (every march step priced as an independent 8-corner access,
                                     # mostly served by L2: above 1) is kept beside it as lookup_rate_*
                                     "frac": (round(committed_traffic("fused-256-512x32")[0] / (avg_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                              if avg_pass and committed_traffic("fused-256-512x32")[0] else None),
                                     "lookup_rate_GBs": round(b_a / (avg_pass * 1e-3) / 1e9, 1) if avg_pass else None,
                                     "lookup_rate_frac": round(b_a / (avg_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg_pass else None,
                                     "algorithmic_bytes": b_a,
                                     # (tile_reduce only sums the volpathsimple half's records since round 5: the nerf half's splats are pre-reduced in LDS by its own kernel)
                                     "secondary": committed_secondary("fused-256-512x32", dict(ca, n_sc=ca["n_sc"] - n_q, n_sc_alb=ca["n_sc_alb"] - n_q)),
                                     # HBM-side bytes of the adjoint pass from the request-size counters (committed profile of
                                     # these kernel sources, or null) and the bandwidth they mean over the measured pass time
                                     "traffic": committed_traffic("fused-256-512x32")[0],
                                     "traffic_GBs": (round(committed_traffic("fused-256-512x32")[0] / (avg_pass * 1e-3) / 1e9, 1)
                                                     if avg_pass and committed_traffic("fused-256-512x32")[0] else None),
                                     "traffic_frac": (round(committed_traffic("fused-256-512x32")[0] / (avg_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                      if avg_pass and committed_traffic("fused-256-512x32")[0] else None)},
                "roofline_primal": {"lookup_rate_GBs": round(b_p / (avg_p * 1e-3) / 1e9, 1) if avg_p else None,
                                    "lookup_rate_frac": round(b_p / (avg_p * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg_p else None,
                                    "algorithmic_bytes": b_p},
                "roofline_note": "SURVEY 8d convention (every lookup / splat priced as an independent 8-corner access); "
                                 "consecutive march steps of one ray share cache lines, so lookup_rate_* is a lookup rate served "
                                 "mostly by L2 and can exceed 1 - `frac` is the counter traffic.  Round 5: the nerf half's splats are "
                                 "pre-reduced in an LDS window (drt_nerf_tile.hip, ds_add_u64 fixed point) and reach the grids as ~1e8 "
                                 "atomics per step instead of 907 M records; the volpathsimple half runs through the production tracers",
                "workload": "config 5 as BASELINE states it: nerf (128 queries) together with volpathsimple-drt over ONE set of grids "
                            "[sigma_t,r,g,b] (emission = albedo), one call per pass, gradients of both into one pair of grids; 256^3, 512x512x32spp"}

    guarded("config2_smoke128_512x16", cfg2)
    guarded("headline_global_majorant" if main_factor else "headline_majorant_factor8", other_factor)
    guarded("headline_envmap_factor8", envmap8)
    guarded("headline_envmap", envmap)
    guarded("config3_optimize_loop", cfg3)
    guarded("config3_as_reproduce", cfg3_reproduce)
    guarded("headline_rank_share", rank_share)
    guarded("config4_512_rank_share_1024x64", cfg4)
    guarded("config5_nerf_256_512x32", cfg5)
    def cfg5_fused_both():
        r = cfg5_fused()
        # the reference's nerf scenes are lit by an environment map and use majorant_resolution_factor 8 (scene_config.py:36,102-141):
        # the same pass in that set-up (the volpathsimple half then tracks through the supergrid on every path's own lane)
        r["envmap_factor8"] = cfg5_fused(env=True, factor=8)
        return r

    guarded("config5_fused_nerf_drt_256_512x32", cfg5_fused_both)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=256, help="grid resolution (headline: 256)")
    ap.add_argument("--film", type=int, default=512, help="film size (headline: 512)")
    ap.add_argument("--spp", type=int, default=32, help="samples per pixel (headline: 32)")
    ap.add_argument("--workload", default="dust-devil", choices=["dust-devil", "smoke", "cube"])
    ap.add_argument("--integrator", default="volpathsimple-drt")
    ap.add_argument("--majorant-factor", type=int, default=8,
                    help="majorant_resolution_factor of the medium (reference scenes: 8, scene_config.py:36; 0 = global majorant)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the other BASELINE configurations (reported under `other_configs`, outside the timed loop)")
    ap.add_argument("--only-config", default=None,
                    help="run ONE entry of other_configs and print it (counter passes of a single configuration under rocprofv3)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU sample (0 = auto)")
    ap.add_argument("--debug-flags", type=int, default=0, help="profiling ablations (drt_set_debug_flags); invalidates the result")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import uivr_amd as u
    from uivr_amd import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the DRT integrator has no CPU path)")
    # DRT_BENCH_SAME_DEVICE=1 (with DRT_BENCH_BACKEND=gloo) lets the multi-rank path be smoke-tested on
    # a single-GPU box; real runs use one GPU per rank over RCCL ("nccl")
    dev_index = 0 if os.environ.get("DRT_BENCH_SAME_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DRT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    if args.only_config:
        print(json.dumps(other_configs(torch, u, synthetic, dev, integ_name=args.integrator, only=args.only_config,
                                       main_factor=args.majorant_factor)), flush=True)
        return

    # ---- workload (synthetic, seeded; resident in HBM) ---------------------------------
    if args.workload == "dust-devil":
        scene = synthetic.dust_devil_scene(res=args.res, film=args.film, device=dev)
    elif args.workload == "smoke":
        scene = synthetic.smoke_scene(res=args.res, film=args.film, device=dev)
    else:
        scene = synthetic.constant_cube_scene(res=args.res, film=args.film, device=dev)
    scene.medium.majorant_resolution_factor = args.majorant_factor
    sensor = scene.sensors[0]
    n_pixels = sensor.width * sensor.height
    spp = args.spp
    n_total = n_pixels * spp
    # --debug-flags (profiling ablations) need the library flavour with test hooks; the default run uses the production one
    integ = u.get_int_config(args.integrator).create(max_depth=int(os.environ.get("DRT_BENCH_MAX_DEPTH", "64")), **({"test_hooks": True} if args.debug_flags else {}))
    shard = u.ShardSpec(rank, world, u.ShardSpec.default_chunk(n_pixels, world)) if world > 1 else None
    batch_shard = shard or u.ShardSpec()
    n_local_pix = batch_shard.n_local_pixels(n_pixels)
    off, inter = batch_shard.ray_mapping(spp)
    batch = u.RayBatch(n_rays=n_local_pix * spp, spp=spp, sensor=sensor, ray_offset=off, interleave=inter)
    grads = u.alloc_grads(scene)
    loss_scale = 2.0 / (n_pixels * 3)
    seed_base = 988378   # opt_config.py:24

    ar_stats = {}

    def step(i):
        seed = u.sample_tea_32(2 * i + 1, seed_base)[0]           # seed_grad of iteration i (optimize.py:328)
        sampler = u.IndependentSampler(seed, spp)
        grads["_flat"].zero_()
        # N > 1: the blocks of the gradient buffer that can be non-zero, from the replicated sigma_t, BEFORE the passes are
        # enqueued (as render_backward does; an optimisation's sigma_t changes every step, so it is part of the step)
        support = u.gradient_support(scene.medium.sigma_t, grads) if world > 1 else None
        L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)       # batched.py:255-264
        img = integ.develop(scene, L, spp)                                                # :272-297
        grad_img = loss_scale * (img - 0.5)                                               # d mean((img-.5)^2)
        dL = integ.film_backward(scene, grad_img, spp)                                    # :298-306
        integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)   # :309-318
        u.allreduce_gradients(grads, stats=ar_stats, support=support)                     # ONE RCCL all-reduce (blocks that can be non-zero), no host wait
        return img

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    h = integ.native_handle(scene)
    if args.debug_flags:
        h.set_debug_flags(args.debug_flags)
    for i in range(args.warmup):
        step(i)
    sync()
    h.enable_timing(True)       # HIP event pairs around every tracing launch, on the launch stream
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync()
    elapsed = time.perf_counter() - t0
    u.verify_pending()                               # (the deferred check of the last step's gradient all-reduce)
    t_primal = h.read_timings(0)
    t_adjoint = h.read_timings(1)
    t_untile = h.read_timings(2)
    t_pass = h.read_timings(3)
    h.enable_timing(False)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_total * args.steps / elapsed / 1e6

    # ---- event counts of ONE step (deterministic; outside the timed region) ------------
    h.enable_counters(True)
    seed_c = u.sample_tea_32(2 * args.warmup + 1, seed_base)[0]
    sampler = u.IndependentSampler(seed_c, spp)
    h.reset_counters()
    L, _, state = integ.sample(u.ADMode.Primal, scene, sampler.clone(), batch)
    cnt_p = {k: int(v) for k, v in h.get_counters().items()}
    img = integ.develop(scene, L, spp)
    dL = integ.film_backward(scene, loss_scale * (img - 0.5), spp)
    h.reset_counters()
    grads["_flat"].zero_()
    integ.sample(u.ADMode.Backward, scene, sampler, batch, δL=dL, state_in=state, grads=grads)
    cnt_a = {k: int(v) for k, v in h.get_counters().items()}
    h.enable_counters(False)
    n_local = batch.n_rays
    bytes_p = algorithmic_bytes(cnt_p, n_local, primal_io=True, adjoint_io=False) - 24 * n_local  # rays generated on device
    bytes_a = algorithmic_bytes(cnt_a, n_local, primal_io=False, adjoint_io=True)
    avg_p = sum(t_primal) / max(1, len(t_primal))
    avg_a = sum(t_adjoint) / max(1, len(t_adjoint))
    avg_r = sum(t_untile) / max(1, len(t_untile))
    # the adjoint's splats are finished by the gradient reduction that follows the tracer (record
    # partition + LDS tile reduction, or the apron-scratch reduction): price the pass as a whole
    # (sub-batches are pipelined over two streams, so the pass is timed as a whole on the launch stream;
    # tracer / reduction times are sums over the sub-batch launches of one step)
    per_step = max(1, len(t_pass))
    avg_a = sum(t_adjoint) / per_step
    avg_r = sum(t_untile) / per_step
    avg_pass = sum(t_pass) / per_step
    ach_a = bytes_a / (avg_pass * 1e-3) / 1e9 if avg_pass > 0 else 0.0
    ach_p = bytes_p / (avg_p * 1e-3) / 1e9 if avg_p > 0 else 0.0
    # HBM-side traffic of the adjoint pass: a hardware-counter figure (rocprofv3 --pmc passes, tools/pmc_to_traffic.py) that
    # this run cannot measure itself.  The committed profile is quoted ONLY if it was taken from the very kernel sources
    # this run executes (hash over csrc/ + include/, recorded by tools/pmc_to_traffic.py); otherwise the field is null.
    traffic, traffic_source = committed_traffic(f"{args.workload}-{args.res}-{args.film}x{spp}" +
                                                (f"-factor{args.majorant_factor}" if args.majorant_factor else ""))
    roofline = {
        "bound": "hbm", "kernel": ("adjoint pass (sample(Backward)): " + ("trace_sq_kernel<adjoint>" if args.majorant_factor else "trace_coop_kernel<adjoint>") + " (dominant, sum_tracer_ms) + record partition + tile_reduce"),
        "achieved": round(ach_a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach_a / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
        # what actually binds the pass (SURVEY.md 8d "secondary ceilings"): vector-instruction issue of the tracers (lanes
        # active, VALU-busy) and the LDS-atomic rate of tile_reduce; from the committed counter pass, hash-gated like `traffic`
        "secondary": committed_secondary(f"{args.workload}-{args.res}-{args.film}x{spp}" +
                                         (f"-factor{args.majorant_factor}" if args.majorant_factor else ""), cnt_a),
        "algorithmic_bytes_per_launch": bytes_a, "avg_launch_ms": round(avg_pass, 4),
        "sum_tracer_ms": round(avg_a, 4), "sum_reduction_ms": round(avg_r, 4),
        "bytes_per_sample_h1": round((bytes_p + bytes_a) / n_local, 1),
        "primal": {"achieved": round(ach_p, 2), "frac": round(ach_p / HBM_PEAK_GBS, 5),
                   "algorithmic_bytes_per_launch": bytes_p, "avg_launch_ms": round(avg_p, 4)},
    }

    # ---- CPU baseline: the oracle (a port, NOT the reference's llvm_ad_rgb) ------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        cores = os.cpu_count() or 1
        cpu_scene = u.Scene(medium=u.GridMedium(sigma_t=scene.medium.sigma_t.cpu().numpy(),
                                                albedo=scene.medium.albedo.cpu().numpy(),
                                                bbox_min=scene.medium.bbox_min, bbox_max=scene.medium.bbox_max,
                                                scale=scene.medium.scale,
                                                majorant_resolution_factor=args.majorant_factor),
                            emitter=scene.emitter, sensors=scene.sensors)
        osc = ob.OracleScene(cpu_scene)
        cpu_spp = args.cpu_spp
        if cpu_spp <= 0:
            # calibrate on the full image at 1 spp, then size the sample for ~15 s of CPU work
            tc = time.perf_counter()
            ob.h1_step(osc, integ.props(), 1, seed_c)
            t1 = max(1e-3, time.perf_counter() - tc)
            cpu_spp = int(max(1, min(spp, round(15.0 / t1))))
        tc = time.perf_counter()
        ob.h1_step(osc, integ.props(), cpu_spp, seed_c)
        dt_atomic = time.perf_counter() - tc
        # the same sample with a per-thread write-combining cache (2^16 voxels) in front of the shared gradient
        # grids: the plain port spends most of its time in contended `omp atomic` adds, which says more about
        # atomics than about the algorithm.  The better of the two is the stated baseline; both are reported.
        # (2^16 lines of 40 B per thread, and 2^20 - 40 MB per thread, 10 GB on a 256-thread host: a splat's voxel stays cached for the whole
        #  pixel neighbourhood that hits it)
        # ... and -1: TILE-BINNED accumulation, the CPU counterpart of the device's deferred splatting (oracle/drt_oracle.c: gbucket_t) - every thread
        # appends records to its own bucket of the splat's z layer, the layers are then reduced without atomics in an even and an odd sweep
        dt_cached, cache_log2 = None, None
        for lg in (16, 20, -1):
            tc = time.perf_counter()
            ob.h1_step(osc, integ.props(), cpu_spp, seed_c, grad_cache_log2=lg)
            d = time.perf_counter() - tc
            if dt_cached is None or d < dt_cached:
                dt_cached, cache_log2 = d, lg
        dt = min(dt_atomic, dt_cached)
        tc = time.perf_counter()
        ob.render_primal(osc, integ.props(), cpu_spp, seed_c)                  # the primal pass alone (no gradient grids)
        dt_primal = time.perf_counter() - tc
        cpu_baseline = {
            "value": round(n_pixels * cpu_spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores,
            "kind": "port",
            "sample": f"same workload, full {sensor.width}x{sensor.height} image at {cpu_spp} spp "
                      f"({n_pixels * cpu_spp} samples, {dt:.1f} s); oracle/drt_oracle.c with OpenMP over rays, gradients "
                      + (("tile-binned: per-thread record buckets by z layer, reduced without atomics " if cache_log2 == -1 else
                          f"through a per-thread write-combining cache of 2^{cache_log2} voxels ") if dt_cached <= dt_atomic else "as atomic adds into the shared grids ") +
                      f"(the best of the accumulation modes, the others reported beside it) "
                      f"(the reference's llvm_ad_rgb needs Mitsuba 3 / Dr.Jit, absent here)",
            "value_shared_atomics": round(n_pixels * cpu_spp / dt_atomic / 1e6, 4),
            "value_thread_local_cache": round(n_pixels * cpu_spp / dt_cached / 1e6, 4),
            # where the port's time goes: the primal pass alone runs an order of magnitude faster than the step - the adjoint's
            # ~100 fp64 adds per sample into grids shared by all threads (537 MB at 256^3) bound it, not the tracking
            "primal_pass_only": round(n_pixels * cpu_spp / dt_primal / 1e6, 4),
        }

    # ---- the other BASELINE configurations (outside the timed loop; N = 1 only) ---------
    other = None
    if rank == 0 and world == 1 and not args.no_extra_configs and args.workload == "dust-devil" and args.res == 256:
        grads = L = state = img = dL = None          # free the headline buffers first
        torch.cuda.empty_cache()
        h.release_scratch()
        other = other_configs(torch, u, synthetic, dev, integ_name=args.integrator, main_factor=args.majorant_factor)

    # the side configurations' headline numbers in ONE compact dict inside fields the driver's record keeps (`roofline`, `config`): the
    # full entries stay under `other_configs`
    side = None
    if other:
        def num(path, key="value"):
            d = other
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d.get(key) if isinstance(d, dict) else None

        lv256 = None
        for lv in (num(("config3_as_reproduce",), "levels") or []):
            if lv.get("grid") == "256^3":
                lv256 = lv.get("iterations_per_s")
        side = {"unit": "Msamples/s unless the key says it/s or ms",
                "config2_smoke128_512x16": num(("config2_smoke128_512x16",)),
                "headline_global_majorant": num(("headline_global_majorant",)) or num(("headline_majorant_factor8",)),
                "headline_envmap_factor8": num(("headline_envmap_factor8",)),
                "config3_it_per_s_factor8": num(("config3_optimize_loop",)),
                "config3_it_per_s_global_majorant": num(("config3_optimize_loop", "global_majorant")),
                "config3_it_per_s_envmap_factor8": num(("config3_optimize_loop", "envmap_factor8")),
                "config3_as_reproduce_it_per_s": num(("config3_as_reproduce",)),
                "config3_as_reproduce_it_per_s_at_256": lv256,
                "config4_512_rank_share": num(("config4_512_rank_share_1024x64",)),
                "config5_nerf": num(("config5_nerf_256_512x32",)),
                "config5_fused_nerf_drt": num(("config5_fused_nerf_drt_256_512x32",)),
                "headline_rank_share_G8_ms_UNMEASURED_ON_MULTI_GPU": num(("headline_rank_share", "G8"), "ms_per_step"),
                "config4_G8_step_ms_ring_vs_direct_UNMEASURED_ON_MULTI_GPU": num(("config4_512_rank_share_1024x64", "projection_G8_UNMEASURED_ON_MULTI_GPU"), "step_ms_ring_vs_direct")}
        roofline["side"] = side

    if rank == 0:
        out = {
            "metric": "Msamples/s primal+adjoint DRT, 256^3 grid 512^2x32spp",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload} {args.res}^3 sigma_t+albedo, {sensor.width}x{sensor.height}x{spp}spp, "
                                   f"{args.integrator}, max_depth 64, majorant_resolution_factor {args.majorant_factor} "
                                   f"({'the reference default, scene_config.py:36; the global majorant is reported under other_configs.headline_global_majorant' if args.majorant_factor == 8 else 'global majorant; the reference default 8 is reported under other_configs.headline_majorant_factor8'}), single sensor, "
                                   f"image tiles sharded over {world} GPU(s)",
                       "n_samples_per_step": n_total, "grid": [args.res] * 3, "film": [sensor.width, sensor.height],
                       "spp": spp, "integrator": args.integrator, "side": side},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "t_primal_ms": round(avg_p, 3), "t_adjoint_ms": round(avg_a, 3),
            "t_grad_reduce_ms": round(avg_r, 3),
            "counters_primal": cnt_p, "counters_adjoint": cnt_a,
            "allreduce": ({"mode": ar_stats.get("mode"), "MiB": round(ar_stats.get("floats", 0) * 4 / 2 ** 20, 1),
                           "of_MiB": round(grads["_flat"].numel() * 4 / 2 ** 20, 1) if grads else None,
                           "active_fraction": round(ar_stats.get("active_fraction", 1.0), 4),
                           "collectives_per_backward": ar_stats.get("collectives")} if world > 1 else None),
            "other_configs": other,
        }
        if args.debug_flags:
            out["INVALID_ablation_debug_flags"] = args.debug_flags
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
