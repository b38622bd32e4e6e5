"""Debug aid for tests/test_gpu_fuzz.py::test_random_sequence_on_one_handle_matches_the_oracle: replays a seed, printing every step, and on a mismatch
repeats the step on a FRESH integrator."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import uivr_amd as uivr
from oracle import binding as oracle
import test_gpu_fuzz as F

gpu = torch.device("cuda:0")
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(77_000 + seed)
    c = F._draw(uivr, seed + 2600, medium_size=seed % 5 == 4)
    props = c["props"]
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    scene = c["scene"]; sg = uivr.scene_to(scene, gpu)
    print("seed", seed, props)
    for step in range(8):
        k = int(rng.integers(0, 6)) if step else 0
        if k == 1:
            c2 = F._draw(uivr, int(rng.integers(0, 10_000)) + 5000, medium_size=rng.random() < 0.2)
            scene = c2["scene"]; sg = uivr.scene_to(scene, gpu)
        elif k == 2:
            f = np.float32(rng.random() * 1.5 + 0.25)
            scene.medium.sigma_t = (np.asarray(scene.medium.sigma_t) * f).astype(np.float32); sg.medium.sigma_t.mul_(float(f))
        elif k == 3:
            al = np.asarray(scene.medium.albedo); al2 = np.clip(al * np.float32(0.9) + np.float32(0.03), 0.0, 1.0).astype(np.float32)
            scene.medium.albedo = al2; sg.medium.albedo.copy_(torch.from_numpy(al2))
        elif k == 4:
            em = uivr.EnvmapEmitter(pixels=F._random_map(rng), scale=float(rng.random() + 0.2), to_world=F._random_rotation(rng)) if rng.random() < 0.5 \
                else uivr.ConstantEmitter(tuple(float(v) for v in rng.random(3) + 0.1))
            scene = uivr.Scene(medium=scene.medium, emitter=em, sensors=scene.sensors)
            sg = uivr.Scene(medium=sg.medium, emitter=uivr.scene_to(uivr.Scene(medium=scene.medium, emitter=em, sensors=[]), gpu).emitter, sensors=sg.sensors)
        spp, rs = int(rng.choice([1, 2, 3, 4, 8])), int(rng.integers(1, 2**31 - 1))
        s = scene.sensors[0]; n_pix = s.width * s.height
        m = scene.medium
        osc = oracle.OracleScene(scene)
        ref = oracle.h1_step(osc, props, spp, rs)
        def run(ig):
            img = uivr.render_primal(sg, ig, 0, spp, rs)
            g = uivr.render_backward(sg, ig, ((2.0 / (n_pix * 3)) * (img - 0.5)).contiguous(), 0, spp, rs)
            out = []
            for key, r in ((uivr.SIGMA_T_KEY, ref["grad_sigma_t"]), (uivr.ALBEDO_KEY, ref["grad_albedo"])):
                gg = g[key].double().cpu().numpy().reshape(r.shape)
                out.append((float(np.abs(gg - r).max()), float(np.abs(r).max()), int((gg != 0).sum()), int((r != 0).sum())))
            return out
        res = run(integ)
        bad = any(e > 2e-4 * mx + 1e-9 for e, mx, _, _ in res)
        print(f" step {step} kind {k} grid {tuple(np.asarray(m.sigma_t).shape[:3])} colour {tuple(np.asarray(m.albedo).shape[:3])} factor {m.majorant_resolution_factor} "
              f"emitter {type(scene.emitter).__name__} film {(s.width, s.height)} spp {spp} rays {n_pix * spp}: {'BAD' if bad else 'ok'} {res}")
        if bad:
            fresh = uivr.load_dict(dict(type="volpathsimple", **props))
            print("   fresh integrator:", run(fresh))
            print("   same integrator again:", run(integ))
            break
