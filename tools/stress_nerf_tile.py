#!/usr/bin/env python3
"""Repetition test of the nerf tile adjoint and the fused pass at size (config 5): the same seeds over and over - radiance bitwise the same,
gradients the same up to summation order (window sums are exact integers; the flushes and the volpathsimple half's reduction add floats) and finite.

    python tools/stress_nerf_tile.py [--reps 20]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import torch
    import uivr_amd as u
    from uivr_amd import synthetic
    dev = torch.device("cuda:0")
    bad = 0
    for name, film, spp, res in (("nerf", 512, 32, 256), ("nerf-drt-fused", 512, 32, 256), ("nerf", 97, 5, 64), ("nerf-drt-fused", 70, 19, 48)):
        sc = synthetic.dust_devil_scene(res=res, film=film, device=dev)
        sc.medium.emission = sc.medium.albedo
        integ = u.get_int_config(name).create(max_depth=64)
        ref = None
        for rep in range(a.reps):
            img = u.render_primal(sc, integ, 0, spp, 77)
            g = u.render_backward(sc, integ, ((2.0 / img.numel()) * (img - 0.4)).contiguous(), 0, spp, 77)["_flat"]
            torch.cuda.synchronize()
            if ref is None:
                ref = (img.clone(), g.clone())
                continue
            same = torch.equal(img, ref[0]); fin = bool(torch.isfinite(g).all())
            err = float((g - ref[1]).abs().max() / ref[1].abs().max())
            if not same or not fin or err > 2e-5:
                bad += 1
                print(f"MISMATCH {name} film {film} rep {rep}: image same {same}, finite {fin}, grad err {err:.2e}")
        print(f"{name}: film {film}^2 x {spp} spp, {res}^3: {a.reps} repetitions done")
    print("STRESS_OK" if bad == 0 else f"STRESS_FAILED {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
