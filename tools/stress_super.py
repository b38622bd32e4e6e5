#!/usr/bin/env python3
"""Stress of the supergrid tracer's LDS protocol (drt_super.hip): the headline scene at majorant_resolution_factor 8, the same
seed over and over - radiance must be bitwise the same every time (a lost or duplicated flight, a stale slot or a missed
done bit would change rays), gradients the same up to summation order and finite; a few seeds and film sizes so that the
end-of-kernel phase (few rays left per wave) is hit in different shapes.

    python tools/stress_super.py [--reps 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import uivr_amd as u
    from uivr_amd import synthetic
    dev = torch.device("cuda:0")
    bad = 0
    for film, spp, res, factor in ((512, 32, 256, 8), (200, 7, 128, 8), (96, 3, 64, 4), (33, 5, 48, 3)):
        scene = synthetic.dust_devil_scene(res=res, film=film, device=dev)
        scene.medium.majorant_resolution_factor = factor
        integ = u.get_int_config("volpathsimple-drt").create(max_depth=64)
        ref_img = ref_g = None
        for rep in range(args.reps):
            seed = 1000 + (rep % 3)
            img = u.render_primal(scene, integ, 0, spp, seed)
            g = u.render_backward(scene, integ, ((2.0 / img.numel()) * (img - 0.5)).contiguous(), 0, spp, seed)["_flat"]
            torch.cuda.synchronize()
            if rep < 3:
                if rep == 0:
                    ref_img, ref_g = {}, {}
                ref_img[seed], ref_g[seed] = img.clone(), g.clone()
                continue
            same = torch.equal(img, ref_img[seed])
            fin = bool(torch.isfinite(g).all())
            err = float((g - ref_g[seed]).abs().max() / ref_g[seed].abs().max())
            if not same or not fin or err > 2e-5:
                bad += 1
                print(f"MISMATCH film {film} rep {rep}: image same {same}, finite {fin}, grad err {err:.2e}")
        print(f"film {film}^2 x {spp} spp, {res}^3 at factor {factor}: {args.reps} repetitions done")
    print("STRESS_OK" if bad == 0 else f"STRESS_FAILED {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
