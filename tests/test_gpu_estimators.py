"""What pins the gradient estimators when no Mitsuba output can: statistics at sample counts only
the GPU affords.

1. `test_reference_test04_protocol_full_width`: the reference's own gradient protocol
   (tests/test_integrators.py:261-347 + python/fd.py) at full width - all 27 sigma_t + 81 albedo entries of the 3^3
   fixture, 128^2 film, eps = 5e-3, loss mean((img - 0.5)^2), AD at 512 spp - for EVERY estimator, judged by the
   reference's own (disabled, `if False:`) thresholds: per parameter and channel at most 3 entries off by more than
   rtol 3e-2, none off by more than rtol 0.75 - counting a deviation only when it is also significant at 4 standard
   errors (both sides are Monte Carlo estimates).  Finite differences are central at 128^2 x 32768 spp, two seeds
   (the reference's forward differences at 4096 spp are too noisy for sigma_t, where a perturbation flips
   real / null decisions); the AD side is the mean of 8 runs at the reference's 512 spp.

2. `test_drt_subsampling_bias_is_the_rgb_mean_reservoir`: quantifies DESIGN.md's finding.  `volpathsimple-drt`
   (`use_drt_subsampling`) is biased when the path throughput is coloured: `DRTReservoir.update` accepts with
   probability mean_rgb(w / wsum) and `get()` returns mean(wsum) * w / mean(w) (volpathsimple.py:751, 756-760) -
   a mean of ratios where an unbiased size-1 reservoir needs the ratio of means.  With a grey albedo every estimator
   agrees with every other (|z| < 5 over all 108 entries); with the fixture's coloured albedo the two subsampling
   estimators - and only they - are off by tens of standard errors, while `quadratic` (same E2 sampler, no reservoir)
   and `basic` still agree.  So E2 (`sample_interaction_drt`) is NOT what is off (its direct KAT:
   tests/test_oracle_e2.py); the reservoir is, and it is the reference's arithmetic, reproduced on purpose.
   This is also why the reference's test_04 (which runs `use_drt_subsampling=True`) has its asserts switched off.
"""
import numpy as np
import pytest
import torch

from conftest import VARIANTS, props_for

pytestmark = pytest.mark.gpu

GROUPS = {"sigma_t": slice(0, 27), "albedo_r": slice(27, 108, 3), "albedo_g": slice(28, 108, 3), "albedo_b": slice(29, 108, 3)}


def _integrator(uivr, variant):
    d = {"type": "volpathsimple"}
    d.update(props_for(variant))
    return uivr.load_dict(d)


def _scene(uivr, gpu, film, grey):
    scene = uivr.cube_test_scene(film, film, density_scale=2.0)       # tests/test_integrators.py:19-116
    if grey:
        scene.medium.albedo[...] = scene.medium.albedo.mean(axis=-1, keepdims=True)
    return uivr.scene_to(scene, gpu)


def _h1(uivr, sg, integ, spp, seed):
    img = uivr.render_primal(sg, integ, 0, spp, seed)
    g = uivr.render_backward(sg, integ, ((2.0 / img.numel()) * (img - 0.5)).contiguous(), 0, spp, seed)
    return torch.cat([g[uivr.SIGMA_T_KEY].reshape(-1), g[uivr.ALBEDO_KEY].reshape(-1)]).double().cpu().numpy()


_FD_CACHE = {}


def _fd_central(uivr, gpu, grey):
    """python/fd.py with central differences: eps 5e-3, the same seed for every render of one pass; two passes
    with different seeds give the estimate (their mean) and its standard error (half their difference)."""
    if grey in _FD_CACHE:
        return _FD_CACHE[grey]
    sg = _scene(uivr, gpu, 128, grey)
    integ = _integrator(uivr, "basic")            # the primal is the same for every estimator
    eps, spp = 5e-3, 32768

    def loss(seed):
        img = uivr.render_primal(sg, integ, 0, spp, seed)
        return float(((img.double() - 0.5) ** 2).mean())

    passes = []
    for seed in (777, 4242):
        out = []
        for t in (sg.medium.sigma_t, sg.medium.albedo):
            flat = t.view(-1)
            for i in range(flat.numel()):
                orig = float(flat[i])
                flat[i] = orig + eps
                lp = loss(seed)
                flat[i] = orig - eps
                lm = loss(seed)
                flat[i] = orig
                out.append((lp - lm) / (2 * eps))
        passes.append(np.array(out))
    fd = 0.5 * (passes[0] + passes[1])
    se = 0.5 * np.abs(passes[0] - passes[1])
    for sl in GROUPS.values():                    # a 2-sample error estimate can be accidentally tiny: floor it per group
        se[sl] = np.maximum(se[sl], np.median(se[sl]))
    _FD_CACHE[grey] = (fd, se, sg.medium.albedo.reshape(-1).cpu().numpy().copy())
    return _FD_CACHE[grey]


def _protocol(a, b, se, mask=None):
    """The reference's criteria (tests/test_integrators.py:324-347) per parameter / channel: number of entries with
    |a - b| >= rtol 3e-2 * |b| ("bad"; at most 3 allowed) and np.allclose(a, b, rtol=0.75).  An entry only counts as
    bad if the deviation is also statistically significant (> 4 standard errors of a - b): with Monte Carlo
    estimates on both sides a 3 % band is inside the noise for the smallest entries even at these sample counts."""
    res = {}
    for name, sl in GROUPS.items():
        aa, bb, ss = a[sl], b[sl], se[sl]
        if mask is not None:
            aa, bb, ss = aa[mask[sl]], bb[mask[sl]], ss[mask[sl]]
        dev = np.abs(aa - bb)
        res[name] = (int(np.sum((dev >= 3e-2 * np.abs(bb)) & (dev > 4.0 * ss))), bool(np.all(dev <= 0.75 * np.abs(bb) + 4.0 * ss)))
    return res


@pytest.mark.parametrize("grey", [False, True], ids=["coloured", "grey"])
def test_reference_test04_protocol_full_width(uivr, gpu, grey):
    fd, fd_se, albedo = _fd_central(uivr, gpu, grey)
    assert np.isfinite(fd).all() and np.abs(fd[:27]).min() > 0
    sg = _scene(uivr, gpu, 128, grey)
    # entries whose albedo is exactly 0 (the fixture's green channel on the z = 2 slab): the free-flight estimator
    # divides the future radiance by max(1e-8, albedo) (volpathsimple.py:167) and is wrong there - documented, kept
    positive = np.concatenate([np.ones(27, bool), albedo > 0])
    results = {}
    for variant in VARIANTS:
        integ = _integrator(uivr, variant)
        runs = np.array([_h1(uivr, sg, integ, 512, 12345 + r) for r in range(8)])           # 512 spp as in the reference
        ad, ad_se = runs.mean(0), runs.std(0, ddof=1) / np.sqrt(runs.shape[0])
        se = np.sqrt(ad_se ** 2 + fd_se ** 2)
        results[variant] = (_protocol(ad, fd, se), _protocol(ad, fd, se, positive))
    print({k: v[0] for k, v in results.items()})

    def passes(res):
        return all(bad <= 3 and close for bad, close in res.values())

    # the estimator without reservoir and without the albedo division passes the reference's thresholds everywhere
    assert passes(results["quadratic-nomis"][0]), results["quadratic-nomis"]
    # estimators that use the free-flight term pass wherever albedo > 0
    for v in ("basic", "quadratic"):
        assert passes(results[v][1]), (v, results[v])
    if grey:
        # grey throughput: the subsampling reservoir is exact, every estimator passes (albedo > 0 everywhere here)
        assert (albedo > 0).all()
        for v in VARIANTS:
            assert passes(results[v][0]), (v, results[v])
    else:
        # coloured throughput: the reference's test_04 configuration (`drt-nomis`) and `drt` miss the thresholds
        # in several groups - the RGB-mean reservoir (see the module docstring), not noise: the other three pass
        for v in ("drt", "drt-nomis"):
            failing = [g for g, (bad, close) in results[v][1].items() if bad > 3 or not close]
            assert len(failing) >= 2, (v, results[v])


def test_drt_subsampling_bias_is_the_rgb_mean_reservoir(uivr, gpu):
    stats = {}
    for grey in (False, True):
        scene = uivr.cube_test_scene(32, 32, density_scale=2.0)
        scene.medium.albedo[...] = np.clip(scene.medium.albedo, 0.05, 1.0)     # keep `basic` valid (albedo > 0)
        if grey:
            scene.medium.albedo[...] = scene.medium.albedo.mean(axis=-1, keepdims=True)
        sg = uivr.scene_to(scene, gpu)
        for v in VARIANTS:
            integ = _integrator(uivr, v)
            runs = np.array([_h1(uivr, sg, integ, 2048, 1000 + r) for r in range(24)])
            stats[(grey, v)] = (runs.mean(0), runs.std(0, ddof=1) / np.sqrt(runs.shape[0]))
    report = {}
    for grey in (False, True):
        ref_m, ref_s = stats[(grey, "quadratic-nomis")]
        for v in VARIANTS:
            if v == "quadratic-nomis":
                continue
            m, s = stats[(grey, v)]
            z = (m - ref_m) / np.sqrt(s * s + ref_s * ref_s)
            rel = np.abs(m - ref_m)[:27] / np.abs(ref_m[:27])
            report[(grey, v)] = (float(np.abs(z[:27]).max()), float(np.abs(z[27:]).max()), float(np.sqrt((z ** 2).mean())), float(rel.max()))
    print(report)
    for v in ("drt", "drt-nomis", "quadratic", "basic"):
        zs, za, rms, _ = report[(True, v)]
        assert zs < 5 and za < 5.5 and rms < 1.6, ("grey", v, report[(True, v)])          # all agree in expectation
    for v in ("quadratic", "basic"):
        zs, za, rms, _ = report[(False, v)]
        assert zs < 5 and za < 5.5 and rms < 1.6, ("coloured", v, report[(False, v)])
    for v in ("drt", "drt-nomis"):
        zs, za, rms, rel = report[(False, v)]
        assert zs > 10 and za > 20 and rms > 5, ("coloured", v, report[(False, v)])     # measured: 22 / 81 / 30 and 45 / 130 / 57
        assert 0.03 < rel < 0.5, rel                                                      # up to 11 % / 22 % of the sigma_t gradient
