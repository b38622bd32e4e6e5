#!/usr/bin/env python3
"""Exploration behind tests/test_gpu_estimators.py (run on the GPU box, prints JSON):
  A. DRT-subsampling bias: gradients of every estimator on the reference's 3^3 fixture with its coloured albedo
     and with a grey one (channel mean); z-scores of each estimator against `quadratic-nomis` (unbiased, no reservoir).
  B. the reference's test_04 protocol at full width (tests/test_integrators.py:261-347): forward differences
     (fd.py: eps 5e-3, 128^2 x 4096 spp, one seed) for all 27 + 81 entries vs the AD gradient at 512 spp.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import uivr_amd as u
from conftest import VARIANTS, props_for

dev = torch.device("cuda:0")


def integrator(v):
    d = {"type": "volpathsimple"}
    d.update(props_for(v))
    return u.load_dict(d)


def h1(sg, integ, spp, seed):
    img = u.render_primal(sg, integ, 0, spp, seed)
    g = u.render_backward(sg, integ, ((2.0 / img.numel()) * (img - 0.5)).contiguous(), 0, spp, seed)
    return torch.cat([g[u.SIGMA_T_KEY].reshape(-1), g[u.ALBEDO_KEY].reshape(-1)]).double().cpu().numpy()


def part_a(out):
    for colour in ("coloured", "grey"):
        scene = u.cube_test_scene(32, 32, density_scale=2.0)
        scene.medium.albedo[...] = np.clip(scene.medium.albedo, 0.05, 1.0)
        if colour == "grey":
            scene.medium.albedo[...] = scene.medium.albedo.mean(axis=-1, keepdims=True)
        sg = u.scene_to(scene, dev)
        stats = {}
        for v in VARIANTS:
            integ = integrator(v)
            runs = np.array([h1(sg, integ, 2048, 1000 + r) for r in range(24)])
            stats[v] = (runs.mean(0), runs.std(0, ddof=1) / np.sqrt(runs.shape[0]))
        ref_m, ref_s = stats["quadratic-nomis"]
        for v in VARIANTS:
            m, s = stats[v]
            z = (m - ref_m) / np.sqrt(s * s + ref_s * ref_s + 1e-300)
            rel = np.abs(m - ref_m) / (np.abs(ref_m) + 1e-12)
            out[f"A/{colour}/{v}"] = dict(max_abs_z_sigma=float(np.abs(z[:27]).max()), max_abs_z_albedo=float(np.abs(z[27:]).max()),
                                          mean_z_sigma=float(z[:27].mean()), max_rel_sigma=float(rel[:27].max()),
                                          median_rel_sigma=float(np.median(rel[:27])),
                                          rms_z_sigma=float(np.sqrt((z[:27] ** 2).mean())), rms_z_albedo=float(np.sqrt((z[27:] ** 2).mean())))
            print(colour, v, out[f"A/{colour}/{v}"], flush=True)


def part_b(out):
    scene = u.cube_test_scene(128, 128, density_scale=2.0)
    sg = u.scene_to(scene, dev)
    eps = 5e-3

    def loss(integ, spp, seed):
        img = u.render_primal(sg, integ, 0, spp, seed)
        return float(((img.double() - 0.5) ** 2).mean())

    integ0 = integrator("drt-nomis")          # the reference's test_04 settings (use_drt_mis False)
    grids = {u.SIGMA_T_KEY: sg.medium.sigma_t, u.ALBEDO_KEY: sg.medium.albedo}
    fds = {}
    for tag, spp, seed in (("ref", 4096, 12345), ("ref_seed2", 4096, 54321), ("hi", 32768, 777)):
        centre = loss(integ0, spp, seed)
        fwd, ctr = [], []
        for key in (u.SIGMA_T_KEY, u.ALBEDO_KEY):
            t = grids[key]
            flat = t.view(-1)
            for i in range(flat.numel()):
                orig = float(flat[i])
                flat[i] = orig + eps
                lp = loss(integ0, spp, seed)
                if tag == "hi":
                    flat[i] = orig - eps
                    lm = loss(integ0, spp, seed)
                    ctr.append((lp - lm) / (2 * eps))
                flat[i] = orig
                fwd.append((lp - centre) / eps)
        fds[tag] = np.array(fwd)
        if ctr:
            fds[tag + "_central"] = np.array(ctr)
        print("fd", tag, "done", flush=True)
    out["B/fd_ref"] = fds["ref"].tolist()
    out["B/fd_ref_seed2"] = fds["ref_seed2"].tolist()
    out["B/fd_hi_forward"] = fds["hi"].tolist()
    out["B/fd_hi_central"] = fds["hi_central"].tolist()
    for v in VARIANTS:
        integ = integrator(v)
        single = h1(sg, integ, 512, 12345)
        many = np.array([h1(sg, integ, 512, 2000 + r) for r in range(16)])
        out[f"B/ad_single/{v}"] = single.tolist()
        out[f"B/ad_mean/{v}"] = many.mean(0).tolist()
        out[f"B/ad_se/{v}"] = (many.std(0, ddof=1) / 4.0).tolist()
        for fd_tag in ("ref", "hi_central"):
            b = fds[fd_tag]
            for name, a in (("single", single), ("mean16", many.mean(0))):
                groups = {"sigma_t": slice(0, 27), "albedo_r": slice(27, 108, 3), "albedo_g": slice(28, 108, 3), "albedo_b": slice(29, 108, 3)}
                res = {}
                for gname, sl in groups.items():
                    aa, bb = a[sl], b[sl]
                    bad = int(np.sum(np.abs(aa - bb) >= 3e-2 * np.abs(bb)))
                    res[gname] = dict(bad=bad, allclose075=bool(np.allclose(aa, bb, rtol=0.75)),
                                      median_rel=float(np.median(np.abs(aa - bb) / (np.abs(bb) + 1e-30))))
                out[f"B/protocol/{v}/{name}_vs_fd_{fd_tag}"] = res
                print(v, name, fd_tag, res, flush=True)


if __name__ == "__main__":
    out = {}
    part_a(out)
    part_b(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "explore_estimators.json"), "w") as f:
        json.dump(out, f)
