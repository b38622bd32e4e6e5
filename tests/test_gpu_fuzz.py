"""Seeded random scenes: the HIP path against the oracle where no hand-picked fixture looks.  Every seed draws its own grid shape (non-cubic,
odd extents, down to 2 voxels per axis), box, medium scale, film shape, sensor (outside or INSIDE the box), emitter (constant / envmap,
with and without hide_emitters), estimator, majorant_resolution_factor, max_depth / rr_depth, spp - and, now and then, a colour grid on its
own lattice or an explicit ray batch.  Bars as everywhere: radiance bit-exact, counters equal, gradients within 2e-4 max|oracle|."""
import os

import numpy as np
import pytest

from conftest import VARIANTS, props_for
from test_oracle_envmap import _blob_map

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4
# (a longer hunt: DRT_FUZZ_SEEDS=lo:hi python -m pytest tests/test_gpu_fuzz.py -m gpu - tools/gpu/run.sh fuzz:lo:hi)
_lo, _hi = (int(v) for v in os.environ.get("DRT_FUZZ_SEEDS", "0:48").split(":"))
SEEDS = list(range(_lo, _hi))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _close(g_hip, g_ref, what):
    g = g_hip.detach().double().cpu().numpy().reshape(g_ref.shape)
    tol = GRAD_RTOL * np.abs(g_ref).max() + 1e-9
    assert np.abs(g - g_ref).max() <= tol, f"{what}: {np.abs(g - g_ref).max():.3e} > {tol:.3e}"


def _draw(uivr, seed):
    rng = np.random.default_rng(10_000 + seed)
    shape = tuple(int(v) for v in rng.integers(2, 29, size=3))                     # (z, y, x)
    kind = rng.integers(0, 4)
    st = rng.random(shape + (1,), dtype=np.float32)
    if kind == 0:
        st = st ** 3 * 8.0                                                          # spiky
    elif kind == 1:
        st = st * 0.3                                                               # thin: most flights leave the box
    elif kind == 2:
        st = st * 4.0
        st[rng.random(shape) < 0.6] = 0.0                                           # mostly empty (occupancy mask, empty supergrid cells)
    else:
        st = st * 0.0 + float(rng.random() * 3.0 + 0.1)                             # homogeneous
    st = st.astype(np.float32)
    own = shape[0] > 3 and rng.random() < 0.25                                      # the colour grid on its own lattice
    cshape = tuple(int(v) for v in rng.integers(2, 20, size=3)) if own else shape
    al = (rng.random(cshape + (3,), dtype=np.float32) * 0.9 + 0.05).astype(np.float32)
    ext = rng.random(3) * 2.5 + 0.5
    centre = rng.normal(size=3) * 0.5
    factor = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 8]))
    medium = uivr.GridMedium(sigma_t=st, albedo=al, bbox_min=tuple(centre - ext / 2), bbox_max=tuple(centre + ext / 2),
                             scale=float(rng.random() * 3.0 + 0.2), majorant_resolution_factor=factor)
    inside = rng.random() < 0.05               # (a ray that starts inside the box neither enters nor escapes: reach_medium, volpathsimple.py:298-310 - radiance 0)
    if inside:
        origin = centre + (rng.random(3) - 0.5) * ext * 0.6
        target = origin + rng.normal(size=3)
    else:
        dirn = rng.normal(size=3)
        dirn /= np.linalg.norm(dirn)
        origin = centre + dirn * (np.linalg.norm(ext) * (0.7 + rng.random() * 2.0))
        target = centre + (rng.random(3) - 0.5) * ext * 0.5
    if abs(np.dot((target - origin) / np.linalg.norm(target - origin), (0.0, 1.0, 0.0))) > 0.98:
        target = target + np.array([0.3, 0.0, 0.2])                                 # (look_at needs a direction that is not the up vector)
    w, h = int(rng.integers(1, 41)), int(rng.integers(1, 41))
    sensor = uivr.PerspectiveSensor(origin=tuple(origin), target=tuple(target), fov=float(rng.random() * 60.0 + 15.0), width=w, height=h)
    env = rng.random() < 0.4
    if env:
        em = uivr.EnvmapEmitter(pixels=_blob_map(), scale=float(rng.random() + 0.2), to_world=uivr.EnvmapEmitter.rotation_y(float(rng.random() * 360.0)))
    else:
        em = uivr.ConstantEmitter(tuple(float(v) for v in rng.random(3) * 1.5 + 0.05))
    variant = list(VARIANTS)[int(rng.integers(0, len(VARIANTS)))]
    max_depth = int(rng.choice([1, 2, 3, 8, 64]))
    over = dict(max_depth=max_depth)
    if rng.random() < 0.3:
        over["rr_depth"] = int(rng.integers(1, 6))                                  # Russian roulette on
    if env and rng.random() < 0.5:
        over["hide_emitters"] = True
    props = props_for(variant, **over)
    spp = int(rng.choice([1, 2, 3, 4, 8, 16]))
    explicit = rng.random() < 0.2
    return dict(scene=uivr.Scene(medium=medium, emitter=em, sensors=[sensor]), props=props, spp=spp, seed=int(rng.integers(1, 2**31 - 1)),
                explicit=explicit, rng=rng, centre=centre, ext=ext, variant=variant, factor=factor, shape=shape, cshape=cshape, film=(w, h), env=env)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_scene_matches_the_oracle(uivr, oracle, gpu, seed):
    c = _draw(uivr, seed)
    scene, props, spp, rs = c["scene"], c["props"], c["spp"], c["seed"]
    tag = f"seed {seed}: {c['variant']} factor {c['factor']} grid {c['shape']} colour {c['cshape']} film {c['film']} spp {spp} env {c['env']} " \
          f"depth {props['max_depth']} rr {props['rr_depth']} explicit {c['explicit']}"
    import torch
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    h = integ.native_handle(sg)
    h.enable_counters(True)
    h.reset_counters()
    if c["explicit"]:
        rng = c["rng"]
        n = int(rng.integers(1, 1500))
        o = (c["centre"] + rng.normal(size=(n, 3)) * np.linalg.norm(c["ext"])).astype(np.float32)
        tgt = (c["centre"] + (rng.random((n, 3)) - 0.5) * c["ext"] * 1.3).astype(np.float32)     # (some rays miss the box)
        d = tgt - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        osc = oracle.OracleScene(scene, sensor_index=None)
        Lr, c_p = oracle.render_primal(osc, props, spp, rs, rays_o=o, rays_d=d)
        dL = ((rng.random((n, 3), dtype=np.float32) - 0.5) * 0.1).astype(np.float32)
        gs, ga, c_a = oracle.render_backward(osc, props, spp, rs, dL, Lr, rays_o=o, rays_d=d)
        batch = uivr.RayBatch(n_rays=n, spp=spp, o=torch.from_numpy(o).to(gpu), d=torch.from_numpy(d).to(gpu))
        samp = uivr.IndependentSampler(rs, spp)
        L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
        np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(Lr), err_msg=tag)
        grads = uivr.alloc_grads(sg)
        integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(gpu), state_in=st, grads=grads)
        expect = {k: c_p[k] + c_a[k] for k in c_p}
    else:
        s = scene.sensors[0]
        n_pix = s.width * s.height
        osc = oracle.OracleScene(scene)
        ref = oracle.h1_step(osc, props, spp, rs)
        _, c_p = oracle.render_primal(osc, props, spp, rs)
        gs, ga = ref["grad_sigma_t"], ref["grad_albedo"]
        batch = uivr.RayBatch(n_rays=n_pix * spp, spp=spp, sensor=sg.sensors[0])
        L, _, _ = integ.sample(uivr.ADMode.Primal, sg, uivr.IndependentSampler(rs, spp), batch)
        np.testing.assert_array_equal(_bits(L.cpu().numpy()), _bits(ref["L"]), err_msg=tag)
        img = uivr.render_primal(sg, integ, 0, spp, rs)
        grads = uivr.render_backward(sg, integ, ((2.0 / (n_pix * 3)) * (img - 0.5)).contiguous(), 0, spp, rs)
        expect = {k: ref["counters"][k] + 2 * c_p[k] for k in ref["counters"]}
    cnt = {k: int(v) for k, v in h.get_counters().items()}
    h.enable_counters(False)
    assert cnt == expect, tag
    _close(grads[uivr.SIGMA_T_KEY], gs, tag + " grad sigma_t")
    _close(grads[uivr.ALBEDO_KEY], ga, tag + " grad albedo")
