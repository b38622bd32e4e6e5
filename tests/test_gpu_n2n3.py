"""GPU-side parity of the rows either side of the hot path and of the host logic that
decides WHICH grid the kernels see:
  * device `upsample_grid` vs `scipy.ndimage.zoom(order=1, mode='nearest', grid_mode=True)` (optimize.py:217-223)
  * device Adam / projection vs the numpy formula (opt_config.py:46-48, optimize.py:169-179, 352-353)
  * `write_vol` bytes of a device grid vs a header assembled independently here (util.py:55-71)
  * the medium binding follows the tensor that is passed, never a recycled address
  * `run_optimization` with parameter-key subsets / supersets (the reference lists sigma_t, albedo, emission)
"""
import struct

import numpy as np
import pytest
import torch
from scipy.ndimage import zoom

from conftest import props_for

pytestmark = pytest.mark.gpu


def test_device_upsample_equals_scipy_zoom(uivr, gpu):
    rng = np.random.default_rng(10)
    for shape in [(4, 6, 5, 1), (3, 3, 3, 3), (16, 8, 12, 3), (32, 32, 32, 1)]:
        a = rng.random(shape, dtype=np.float32)
        new = tuple(2 * s for s in shape[:3]) + (shape[3],)
        ref = zoom(a, [2, 2, 2, 1], order=1, mode='nearest', prefilter=False, grid_mode=True)
        got = uivr.upsample_grid(torch.from_numpy(a).to(gpu), new)
        assert got.is_cuda and got.is_contiguous() and tuple(got.shape) == new
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-6)
    # non-integer factor (a fixed grid resampled next to optimised ones): 6 -> 9
    a = rng.random((6, 6, 6, 3), dtype=np.float32)
    ref = zoom(a, [1.5, 1.5, 1.5, 1], order=1, mode='nearest', prefilter=False, grid_mode=True)
    got = uivr.upsample_grid(torch.from_numpy(a).to(gpu), (9, 9, 9, 3))
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-6)


def test_device_adam_and_projection_equal_formula(uivr, gpu):
    rng = np.random.default_rng(11)
    shape = (8, 8, 8, 3)
    p0 = (rng.random(shape) * 2).astype(np.float32)
    keys = ("medium1.sigma_t.data", "medium1.albedo.data")
    params = {k: torch.from_numpy(p0.copy()).to(gpu) for k in keys}
    opt = uivr.Adam(lr=5e-2, params=params)
    opt.set_learning_rate({keys[1]: 1e-1})
    sc = uivr.SceneConfig(name="c", scene=uivr.cube_test_scene(8, 8), param_keys=list(keys), sensors=[0],
                          start_from_value={k: 0.1 for k in keys}, max_density=1.5)
    ref = {k: p0.astype(np.float64).copy() for k in keys}
    m = {k: np.zeros(shape) for k in keys}
    v = {k: np.zeros(shape) for k in keys}
    lo_hi = {keys[0]: (0.0, 1.5), keys[1]: (0.0, 1.0)}
    lr = {keys[0]: 5e-2, keys[1]: 1e-1}
    for t in range(1, 8):
        g = rng.normal(size=shape).astype(np.float32)
        opt.step({k: torch.from_numpy(g).to(gpu) for k in keys})
        uivr.enforce_valid_params(sc, opt)
        for k in keys:
            m[k] = 0.9 * m[k] + 0.1 * g
            v[k] = 0.999 * v[k] + 0.001 * g.astype(np.float64) ** 2
            ref[k] = ref[k] - lr[k] * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m[k] / (np.sqrt(v[k]) + 1e-8)
            ref[k] = np.clip(ref[k], *lo_hi[k])
    for k in keys:
        np.testing.assert_allclose(params[k].cpu().numpy(), ref[k], rtol=2e-5, atol=2e-6)
        assert params[k].is_cuda


def test_vol_bytes_equal_independent_header(uivr, gpu, tmp_path):
    """python/util.py:55-71 -> mi.VolumeGrid.write: 'V','O','L', u8 3, i32 1 (float32), i32 xres, yres, zres,
    i32 channels, 6 x f32 bbox, then the data with x fastest and channels interleaved."""
    rng = np.random.default_rng(12)
    for c in (1, 3):
        z, y, x = 5, 7, 9
        a = rng.random((z, y, x, c), dtype=np.float32)
        path = str(tmp_path / f"dev{c}.vol")
        uivr.write_vol(path, torch.from_numpy(a).to(gpu), (-1.0, -0.5, 0.25), (1.0, 1.5, 3.0))
        expect = b"VOL" + struct.pack("<B", 3) + struct.pack("<i", 1) + struct.pack("<iii", x, y, z) + struct.pack("<i", c) \
            + struct.pack("<6f", -1.0, -0.5, 0.25, 1.0, 1.5, 3.0)
        body = bytearray()
        for k in range(z):
            for j in range(y):
                for i in range(x):
                    for ch in range(c):
                        body += struct.pack("<f", float(a[k, j, i, ch]))
        assert open(path, "rb").read() == expect + bytes(body)
        back, lo, hi = uivr.read_vol(path)
        np.testing.assert_array_equal(back, a)
        assert lo == (-1.0, -0.5, 0.25) and hi == (1.0, 1.5, 3.0)


def test_fresh_grids_are_never_mistaken_for_the_bound_one(uivr, oracle, gpu):
    """A finite-difference loop `render(params={S: st + eps})`, `render(params={S: st - eps})` allocates a new
    grid for each call; the caching allocator hands a just-freed block back at the same address with version 0.
    The derived device state (brick copy, majorant, mask) must follow the tensor that is passed."""
    scene = uivr.cube_test_scene(16, 16, density_scale=2.0)
    props = props_for("basic")
    integ = uivr.load_dict(dict(type="volpathsimple", **props))
    sg = uivr.scene_to(scene, gpu)
    spp, seed = 8, 77
    base = sg.medium.sigma_t

    def render_with(delta):
        st = base + delta                         # fresh tensor, freed at the end of the call
        img = uivr.render(sg, params={uivr.SIGMA_T_KEY: st, uivr.ALBEDO_KEY: sg.medium.albedo}, integrator=integ,
                          spp=spp, seed=seed)
        return img.cpu().numpy()

    imgs = {}
    for delta in (0.5, -0.25, 0.5, 1.0, -0.25):
        torch.cuda.synchronize()
        img = render_with(delta)
        sc = uivr.cube_test_scene(16, 16, density_scale=2.0)
        sc.medium.sigma_t = (sc.medium.sigma_t + np.float32(delta)).astype(np.float32)
        L, _ = oracle.render_primal(oracle.OracleScene(sc), props, spp, seed)
        np.testing.assert_allclose(img, oracle.develop(L, spp), rtol=0, atol=1e-6, err_msg=str(delta))
        imgs.setdefault(delta, img)
        np.testing.assert_array_equal(img, imgs[delta])
    assert not np.array_equal(imgs[0.5], imgs[-0.25])
    # in-place edits of the SAME tensor are seen too (version counter)
    st = base.clone()
    a = uivr.render(sg, params={uivr.SIGMA_T_KEY: st, uivr.ALBEDO_KEY: sg.medium.albedo}, integrator=integ, spp=spp, seed=seed)
    st.mul_(0.25)
    b = uivr.render(sg, params={uivr.SIGMA_T_KEY: st, uivr.ALBEDO_KEY: sg.medium.albedo}, integrator=integ, spp=spp, seed=seed)
    assert not torch.equal(a, b)


@pytest.mark.parametrize("keys,int_name", [(["sigma_t"], "volpathsimple-drt"), (["albedo"], "volpathsimple-drt"),
                                           (["sigma_t", "albedo", "emission"], "volpathsimple-drt"),
                                           (["sigma_t", "albedo"], "nerf"), (["sigma_t", "emission"], "nerf")])
def test_run_optimization_parameter_key_subsets(uivr, gpu, keys, int_name, tmp_path):
    """optimize.py:134-166 / scene_config.py:148: the optimised keys need not equal the integrator's."""
    from uivr_amd import synthetic
    scene = synthetic.smoke_scene(res=16, film=24, device=gpu, optical_side=8.0)
    scene.sensors = synthetic.ring_sensors(3, radius=5.0, height=0.8, fov=30.0, width=24, film_height=24)
    scene.medium.emission = scene.medium.albedo.clone() * 0.5
    full = {"sigma_t": uivr.SIGMA_T_KEY, "albedo": uivr.ALBEDO_KEY, "emission": uivr.EMISSION_KEY}
    pk = [full[k] for k in keys]
    start = {uivr.SIGMA_T_KEY: 0.4, uivr.ALBEDO_KEY: 0.5, uivr.EMISSION_KEY: 0.1}
    sc = uivr.SceneConfig(name="s", scene=scene, param_keys=pk, sensors=[0, 1, 2], start_from_value={k: start[k] for k in pk},
                          max_depth=8, ref_spp=256, max_density=20.0, ref_integrator=int_name)
    oc = uivr.OptimizationConfig("t", spp=2, n_iter=6, lr=2e-2, primal_spp_factor=2, batch_size=256, upsample=[0.5],
                                 checkpoint_initial=False, checkpoint_final=False, checkpoint_stride=2)
    _, params, opt, hist = uivr.run_optimization(str(tmp_path / "out"), oc, sc, int_name)
    assert sorted(params) == sorted(pk) and len(hist) == 6 and np.isfinite(hist).all()
    for k in pk:
        assert tuple(params[k].shape[:3]) == (16, 16, 16)           # started at 8^3, upsampled once
    integ_keys = uivr.get_int_config(int_name).create(max_depth=8).param_keys
    moved = [k for k in pk if float((params[k] - start[k]).abs().max()) > 0]
    assert sorted(moved) == sorted(k for k in pk if k in integ_keys)    # keys the integrator does not read stay put
    # the stride checkpoint is written although checkpoint_initial is off (util.py:57 always creates the directory)
    import os
    assert any(f.startswith("00000002-") for f in os.listdir(tmp_path / "out" / "params"))


def test_fused_adam_step_equals_elementwise_sequence(uivr, gpu):
    """drt_adam_step (one pass over p, g, m, v) against the same update written as torch elementwise ops - the
    sequence the CPU path of `Adam.step` runs - for sizes that are and are not multiples of four; unaligned buffers
    are refused."""
    native = uivr._native.native
    gen = torch.Generator().manual_seed(3)
    for n in (1, 3, 4, 1023, 4099):
        p = torch.rand(n, generator=gen).to(gpu); g = torch.randn(n, generator=gen).to(gpu)
        m = (torch.randn(n, generator=gen) * 0.1).to(gpu); v = (torch.rand(n, generator=gen) * 0.01).to(gpu)
        b1, b2, eps, lr_t = 0.9, 0.999, 1e-8, 3.1e-3
        pe, me, ve = p.clone(), m.clone(), v.clone()
        me.mul_(b1).add_(g, alpha=1 - b1)
        ve.mul_(b2).addcmul_(g, g, value=1 - b2)
        pe.addcdiv_(me, ve.sqrt().add_(eps), value=-lr_t)
        native().adam_step(torch.cuda.current_stream().cuda_stream, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, b1, b2, eps, lr_t)
        # (torch may contract m * b1 + a * g into an fma: the two agree to an ulp or two)
        torch.testing.assert_close(m, me, rtol=1e-6, atol=2e-8)
        torch.testing.assert_close(v, ve, rtol=1e-6, atol=2e-9)
        torch.testing.assert_close(p, pe, rtol=0, atol=2e-7)
    big = torch.zeros(64, device=gpu)
    with pytest.raises(RuntimeError):
        native().adam_step(0, big.data_ptr() + 4, big.data_ptr(), big.data_ptr(), big.data_ptr(), 8, 0.9, 0.999, 1e-8, 1e-3)


def test_clamped_adam_step_equals_step_then_clamp(uivr, gpu):
    """drt_adam_step_clamped (opt.step + enforce_valid_params, optimize.py:352-353, in one pass) is bit-identical to
    drt_adam_step followed by torch's clamp - NaNs stay NaNs, open sides stay open - and `Adam.step(bounds=...)` /
    `run_optimization` use it: the parameters after a step equal the two-pass sequence."""
    native = uivr._native.native
    gen = torch.Generator().manual_seed(7)
    for n, lo, hi in ((4099, 0.0, 1.0), (1023, 0.0, float("inf")), (64, float("-inf"), 0.3)):
        p = (torch.rand(n, generator=gen) * 1.2 - 0.1).to(gpu); g = (torch.randn(n, generator=gen) * 30).to(gpu)
        p[3] = float("nan")
        m = (torch.randn(n, generator=gen) * 0.1).to(gpu); v = (torch.rand(n, generator=gen) * 0.01).to(gpu)
        p2, m2, v2 = p.clone(), m.clone(), v.clone()
        st = torch.cuda.current_stream().cuda_stream
        native().adam_step(st, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.9, 0.999, 1e-8, 5e-2)
        p.clamp_(None if lo == float("-inf") else lo, None if hi == float("inf") else hi)
        native().adam_step_clamped(st, p2.data_ptr(), g.data_ptr(), m2.data_ptr(), v2.data_ptr(), n, 0.9, 0.999, 1e-8, 5e-2, lo, hi)
        assert torch.equal(torch.nan_to_num(p, nan=-7.0), torch.nan_to_num(p2, nan=-7.0)) and bool(torch.isnan(p2[3]))
        assert torch.equal(m, m2) and torch.equal(v, v2)
        assert float(p2[~torch.isnan(p2)].min()) >= float(np.float32(lo)) and float(p2[~torch.isnan(p2)].max()) <= float(np.float32(hi))
    with pytest.raises(RuntimeError):
        native().adam_step_clamped(0, p2.data_ptr(), g.data_ptr(), m2.data_ptr(), v2.data_ptr(), 8, 0.9, 0.999, 1e-8, 1e-3, 1.0, 0.0)
    # the optimizer: one step with bounds == step + enforce_valid_params
    scene = uivr.scene_to(uivr.cube_test_scene(8, 8), gpu)
    keys = (uivr.SIGMA_T_KEY, uivr.ALBEDO_KEY)
    sc = uivr.SceneConfig(name="c", scene=scene, param_keys=list(keys), sensors=[0], start_from_value={keys[0]: 0.1, keys[1]: 0.5})
    pa = {keys[0]: torch.rand(8, 8, 8, 1, device=gpu), keys[1]: torch.rand(8, 8, 8, 3, device=gpu)}
    pb = {k: t.clone() for k, t in pa.items()}
    grads = {k: torch.randn_like(t) * 50 for k, t in pa.items()}
    oa, ob = uivr.Adam(lr=0.5, params=pa), uivr.Adam(lr=0.5, params=pb)
    done = oa.step(grads, bounds=uivr.optimize.param_bounds(sc, keys))
    assert done == set(keys)
    uivr.enforce_valid_params(sc, oa, skip=done)
    assert ob.step(grads) == set()
    uivr.enforce_valid_params(sc, ob)
    for k in keys:
        assert torch.equal(pa[k], pb[k]) and float(pa[k].min()) >= 0.0
    assert float(pa[keys[1]].max()) <= 1.0


def test_adam_step_on_odd_sized_grids(uivr, gpu):
    """3^3 / 5^3 grids (the finite-difference fixtures): gradient views out of `alloc_grads` are 16-byte aligned, so the
    fused kernel takes them; a deliberately misaligned or mismatched gradient falls back to the torch ops - same result."""
    _fused_adam_ok = uivr.optimize._fused_adam_ok
    for res in (3, 5):
        scene = uivr.scene_to(uivr.cube_test_scene(8, 8), gpu)
        scene.medium.sigma_t = torch.rand(res, res, res, 1, device=gpu)
        scene.medium.albedo = torch.rand(res, res, res, 3, device=gpu)
        keys = (uivr.SIGMA_T_KEY, uivr.ALBEDO_KEY)
        params = {keys[0]: scene.medium.sigma_t.clone(), keys[1]: scene.medium.albedo.clone()}
        twin = {k: v.clone() for k, v in params.items()}
        grads = uivr.alloc_grads(scene)
        for k in keys:
            grads[k].copy_(torch.randn_like(grads[k]))
            assert _fused_adam_ok(params[k], grads[k], torch.zeros_like(params[k]), torch.zeros_like(params[k]))
        opt = uivr.Adam(lr=1e-2, params=params)
        opt.step(grads)
        # the same step with gradients at a misaligned offset of another buffer: torch path
        n0, n1 = params[keys[0]].numel(), params[keys[1]].numel()
        buf = torch.zeros(1 + n0 + n1, device=gpu)
        mis = {keys[0]: buf[1:1 + n0].view_as(params[keys[0]]), keys[1]: buf[1 + n0:].view_as(params[keys[1]])}
        for k in keys:
            mis[k].copy_(grads[k])
        assert not _fused_adam_ok(twin[keys[0]], mis[keys[0]], torch.zeros_like(twin[keys[0]]), torch.zeros_like(twin[keys[0]]))
        opt2 = uivr.Adam(lr=1e-2, params=twin)
        opt2.step(mis)
        for k in keys:
            torch.testing.assert_close(params[k], twin[k], rtol=0, atol=3e-7)
