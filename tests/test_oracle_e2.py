"""Direct known-answer test for E2, `Medium::sample_interaction_drt` (call site
python/integrators/volpathsimple.py:549-551: "the sampling probability is T(t')").

E2 is the least-constrained [M3-ext] restatement (the Mitsuba branch is absent) and the gradients of
`volpathsimple-drt` hinge on it.  Its contract, from the call-site comment and the paper's estimator:
the returned position x' = o + t' d and weight W satisfy, for ANY integrable f,

        E[ W f(t') ]  =  integral_0^maxt  T(t) f(t) dt,        T(t) = exp(-integral_0^t sigma_t)

Checked here against a numerical quadrature of the right-hand side for f in {1, t, sigma_t(t)} on the
reference's heterogeneous 3^3 fixture (tests/test_integrators.py:19-116) and on a sparse random grid, with a
global majorant and with a majorant supergrid.  f = sigma_t has a closed form, 1 - T(maxt), which the
quadrature must reproduce as well.  CPU only (oracle hook `drto_sample_interaction_drt`); the GPU test
`tests/test_gpu_e2.py` then pins the HIP implementation to the oracle bit for bit.
"""
import ctypes as C

import numpy as np
import pytest


def _quadrature(oracle, osc, o, d, maxt, n=20000):
    """midpoint rule for the three integrals; sigma_t along the ray from the oracle's own lookup (pinned
    against numpy in tests/test_oracle_primitives.py)."""
    L = oracle.lib()
    ts = (np.arange(n) + 0.5) / n * maxt
    sig = np.array([L.drto_eval_sigma_t(C.byref(osc.medium), (C.c_float * 3)(*(o + t * d))) for t in ts], dtype=np.float64)
    h = maxt / n
    tau_mid = np.cumsum(sig) * h - 0.5 * sig * h            # optical depth at the midpoints
    T = np.exp(-tau_mid)
    return dict(one=float(np.sum(T) * h), t=float(np.sum(T * ts) * h), sigma=float(np.sum(T * sig) * h),
                T_end=float(np.exp(-np.sum(sig) * h)))


def _check(oracle, osc, o, d, n_walks, seed):
    o = np.asarray(o, dtype=np.float32)
    d = np.asarray(d, dtype=np.float32)
    d = (d / np.linalg.norm(d)).astype(np.float32)
    valid, t, W, maxt = oracle.sample_interaction_drt(osc, o, d, seed, n_walks)
    assert maxt > 0
    q = _quadrature(oracle, osc, o.astype(np.float64), d.astype(np.float64), maxt)
    # closed form of the sigma_t integral: 1 - T(maxt)
    assert q["sigma"] == pytest.approx(1.0 - q["T_end"], rel=2e-4)
    L = oracle.lib()
    W = W.astype(np.float64)
    sig_sel = np.array([L.drto_eval_sigma_t(C.byref(osc.medium), (C.c_float * 3)(*(o + tt * d))) if v else 0.0
                        for tt, v in zip(t, valid)], dtype=np.float64)
    tsel = np.where(valid, t, 0.0).astype(np.float64)
    Wv = np.where(valid, W, 0.0)
    est = dict(one=Wv, t=Wv * tsel, sigma=Wv * sig_sel)
    for k in ("one", "t", "sigma"):
        m, se = est[k].mean(), est[k].std(ddof=1) / np.sqrt(n_walks)
        z = (m - q[k]) / se
        assert abs(z) < 4.5, (k, m, q[k], z)
        assert m == pytest.approx(q[k], rel=0.02), (k, m, q[k])
    # every selected position lies on the segment; W is the same for every f (one walk, one weight)
    assert (t[valid] <= maxt).all() and (t[valid] > 0).all()
    # a walk is invalid only if it produced no tentative collision with positive weight: then W == 0
    assert (W[~valid] == 0).all()
    return q


def test_e2_on_the_reference_fixture(oracle, uivr):
    scene = uivr.cube_test_scene(8, 8, density_scale=2.0)
    osc = oracle.OracleScene(scene)
    q = _check(oracle, osc, (-0.45, -0.3, -0.4), (1.0, 0.9, 0.8), 200000, 7)
    assert 0.05 < q["T_end"] < 0.9                      # a non-trivial segment
    _check(oracle, osc, (1.4, 1.2, -0.45), (-0.8, -0.5, 1.0), 200000, 8)


@pytest.mark.parametrize("factor", [0, 4])
def test_e2_sparse_grid_global_and_supergrid(oracle, uivr, factor):
    """Sparse random 16^3 grid: with the supergrid (`majorant_resolution_factor`, scene_config.py:36) the weights
    are T_i / local majorant; the expectation is the same integral."""
    rng = np.random.default_rng(3)
    st = (rng.random((16, 16, 16, 1), dtype=np.float32) ** 3 * 6.0).astype(np.float32)
    st[:, :, 5:9] = 0.0
    medium = uivr.GridMedium(sigma_t=st, albedo=np.full((16, 16, 16, 3), 0.5, np.float32), bbox_min=(-1, -1, -1),
                             bbox_max=(1, 1, 1), scale=1.2, majorant_resolution_factor=factor)
    osc = oracle.OracleScene(uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter(), sensors=[]), sensor_index=None)
    _check(oracle, osc, (-0.95, -0.4, 0.3), (1.0, 0.35, -0.2), 150000, 21)


def test_e2_empty_segment_is_invalid(oracle, uivr):
    """No density => no tentative collisions => invalid, W = 0 (the caller skips the splat, :557-558)."""
    st = np.zeros((4, 4, 4, 1), np.float32)
    medium = uivr.GridMedium(sigma_t=st, albedo=np.zeros((4, 4, 4, 3), np.float32))
    osc = oracle.OracleScene(uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter(), sensors=[]), sensor_index=None)
    valid, t, W, maxt = oracle.sample_interaction_drt(osc, (0.1, 0.5, 0.5), (1, 0, 0), 1, 64)
    assert not valid.any() and (W == 0).all() and np.isinf(t).all()
