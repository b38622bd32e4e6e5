"""Every BASELINE.json configuration at ITS size on the GPU, checked against the CPU oracle.

The oracle cannot trace 10^7..10^8 rays in seconds, but it takes a ray range (`n_rays` / `ray_offset`), and a
ray's radiance depends only on (seed, global ray index, scene).  So each configuration gets
  (1) the FULL-size launch on the full-size grids (the kernels, block maps, record streams, tile partition
      and path cache the production run uses), of which a window of rays through the dense part of the volume
      is compared with the oracle: radiance BIT-EXACT;
  (2) the same window traced alone (primal + adjoint): event counters EQUAL, gradients within
      2e-4 * max|oracle| (fp32 sums in a different order; the oracle accumulates in fp64);
  (3) size-independent properties of the full launch: determinism (bitwise), linearity of the adjoint in dL, energy bound (the mean
      radiance does not exceed the emitter's), every non-zero 256-byte block of the gradient inside `gradient_support(sigma_t)`, and
      "window gradients contained in the full ones": the FULL launch with dL zeroed outside a window gives the gradient of that window
      traced alone;
  (4) a SECOND window per configuration chosen by a seeded draw among the stretches of a row that hold rays that miss the medium's
      box, rays through its thin edge AND rays through thicker parts (round-4 review: no configuration's window is one hand-picked
      row through the dense part).
Workloads (python/reproduce.py:45-59, python/scene_config.py:108-170):
  config 2  smoke plume 128^3 (janga-smoke stand-in), 512^2 x 16 spp
  config 3  dust devil 256^3, 63 sensors, render_batch(32768 px, spp 1024 / spp_grad 16), 3 optimiser iterations
  headline  dust devil 256^3, 512^2 x 32 spp
  config 4  512^3, rank 0's share (ShardSpec(0, 8)) of 1024^2 x 64 spp
  config 5  nerf integrator (128 queries, opt_config.py:162-169) on the 256^3 grids, 512^2 x 32 spp
            (+ the fused 4-channel pass: tests/test_gpu_fused.py)
"""
import numpy as np
import pytest
import torch

from conftest import props_for

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4


def _integrator(uivr, props):
    d = {"type": "volpathsimple"}
    d.update(props)
    return uivr.load_dict(d)


def _cpu_scene(uivr, sg, sensor=0):
    m = sg.medium
    medium = uivr.GridMedium(sigma_t=m.sigma_t.cpu().numpy(), albedo=m.albedo.cpu().numpy(), bbox_min=m.bbox_min,
                             bbox_max=m.bbox_max, scale=m.scale, majorant_resolution_factor=m.majorant_resolution_factor,
                             emission=None if m.emission is None else m.emission.cpu().numpy())
    return uivr.Scene(medium=medium, emitter=sg.emitter, sensors=sg.sensors)


def _close_on_device(g_hip: torch.Tensor, g_ref: np.ndarray, what: str):
    """|hip - oracle| <= 2e-4 max|oracle| + 1e-9, evaluated on the device (the 512^3 grids are GBs)."""
    ref = torch.from_numpy(g_ref).to(g_hip.device)
    tol = GRAD_RTOL * float(ref.abs().max()) + 1e-9
    err = float((g_hip.double() - ref).abs().max())
    assert float(ref.abs().max()) > 0, f"{what}: the oracle's gradient is identically zero (window misses the volume)"
    assert err <= tol, f"{what}: max abs err {err:.3e} > tol {tol:.3e}"


def _window_check(uivr, oracle, sg, integ, props, spp, seed, L_full, local_first, global_first, n, interleave=None,
                  min_lookups_per_ray=2.0):
    """Rays [local_first, local_first + n) of the full launch == global rays [global_first, global_first + n):
    radiance of the full launch bit-exact vs the oracle; then the window alone: counters, gradients."""
    dev = sg.medium.sigma_t.device
    osc = oracle.OracleScene(_cpu_scene(uivr, sg))
    Lr, c_primal = oracle.render_primal(osc, props, spp, seed, n_rays=n, ray_offset=global_first)
    np.testing.assert_array_equal(L_full[local_first:local_first + n].cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    assert c_primal["n_dt"] >= min_lookups_per_ray * n and c_primal["n_dt"] > 0, "the window misses the volume"
    rng = np.random.default_rng(5)
    dL = ((rng.random((n, 3), dtype=np.float32) - 0.5) * 1e-3).astype(np.float32)
    gs, ga, c_adj = oracle.render_backward(osc, props, spp, seed, dL, Lr, n_rays=n, ray_offset=global_first)
    h = integ.native_handle(sg)
    batch = uivr.RayBatch(n_rays=n, spp=spp, sensor=sg.sensors[0], ray_offset=global_first)
    samp = uivr.IndependentSampler(seed, spp)
    h.enable_counters(True)
    h.reset_counters()
    Lw, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    cp = {k: int(v) for k, v in h.get_counters().items()}
    np.testing.assert_array_equal(Lw.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    assert cp == c_primal
    h.reset_counters()
    grads = uivr.alloc_grads(sg)
    integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dL).to(dev), state_in=st, grads=grads)
    ca = {k: int(v) for k, v in h.get_counters().items()}
    h.enable_counters(False)
    assert ca == c_adj
    _close_on_device(grads[uivr.SIGMA_T_KEY], gs, "window grad sigma_t")
    _close_on_device(grads[uivr.ALBEDO_KEY], ga, "window grad albedo")
    grads["_dL"] = dL
    return grads


def _pixel_classes(sg, n_samples=48):
    """Per pixel of sensor 0 (centre ray): 0 = misses the medium's box, 1 = crosses it with optical depth < 0.05 (thin edge),
    2 = thicker.  Nearest-voxel samples of sigma_t along the chord: a classification for choosing windows, it enters no result."""
    s = sg.sensors[0]
    dev = sg.medium.sigma_t.device
    f = s.frame()
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
    py, px = torch.meshgrid(torch.arange(s.height, device=dev), torch.arange(s.width, device=dev), indexing="ij")
    cx = (1.0 - 2.0 * (px.double() + 0.5) / s.width) * float(f["tan_x"])
    cy = (1.0 - 2.0 * (py.double() + 0.5) / s.height) * float(f["tan_y"])
    d = cx[..., None] * t(f["left"]) + cy[..., None] * t(f["up"]) + t(f["dir"])
    d = d / d.norm(dim=-1, keepdim=True)
    o = t(f["origin"])
    bmin, bmax = t(sg.medium.bbox_min), t(sg.medium.bbox_max)
    inv = 1.0 / torch.where(d.abs() < 1e-30, torch.full_like(d, 1e-30), d)
    ta, tb = (bmin - o) * inv, (bmax - o) * inv
    t0 = torch.minimum(ta, tb).amax(dim=-1).clamp_min(0.0)
    t1 = torch.maximum(ta, tb).amin(dim=-1)
    hit = t1 > t0
    grid = sg.medium.sigma_t.reshape(sg.medium.sigma_t.shape[:3]).double() * float(sg.medium.scale)
    rz, ry, rx = grid.shape
    od = torch.zeros_like(t0)
    for j in range(n_samples):
        tt = t0 + (t1 - t0) * ((j + 0.5) / n_samples)
        p = (o + tt[..., None] * d - bmin) / (bmax - bmin)
        ix = (p[..., 0] * rx).long().clamp(0, rx - 1); iy = (p[..., 1] * ry).long().clamp(0, ry - 1); iz = (p[..., 2] * rz).long().clamp(0, rz - 1)
        od += grid[iz, iy, ix]
    od = od * (t1 - t0).clamp_min(0.0) / n_samples
    cls = torch.where(hit, torch.where(od < 0.05, 1, 2), 0)
    return cls.reshape(-1)                                                   # row-major pixels


def _seeded_window(sg, n_pix, draw_seed, ranges=None):
    """First pixel of a window of `n_pix` consecutive pixels (inside one row, inside one of the pixel `ranges`) drawn - seeded - among
    the windows that hold at least n_pix / 8 rays of each kind: missing the box, through the thin edge, through thicker parts (if no
    window holds all three: thin edge + thicker, else missing + thicker)."""
    s = sg.sensors[0]
    cls = _pixel_classes(sg).cpu().numpy()
    ranges = ranges or [(0, cls.size)]
    need = max(1, n_pix // 8)
    cs = [np.concatenate([[0], np.cumsum(cls == k)]) for k in range(3)]
    starts = np.concatenate([np.arange(lo, hi - n_pix + 1, 8) for lo, hi in ranges])
    starts = starts[(starts % s.width) + n_pix <= s.width]
    cands = np.zeros(0, dtype=np.int64)
    for kinds in ((0, 1, 2), (1, 2), (0, 2)):                   # (a scene whose box fills the image has no missing rays: the thin edge then)
        ok = np.ones(starts.size, dtype=bool)
        for k in kinds:
            ok &= (cs[k][starts + n_pix] - cs[k][starts]) >= need
        cands = starts[ok]
        if cands.size:
            break
    assert cands.size > 0, "no window holds rays of different kinds: the scene's silhouette is not in this pixel range"
    return int(cands[np.random.default_rng(draw_seed).integers(cands.size)])


def _masked_full_equals_window(uivr, sg, integ, spp, seed, batch, local_first, n, grads_window, dLw, L_full_state):
    """ "Window gradients contained in the full ones": the FULL launch (its ray order, batches, record streams and tile partition)
    with dL = 0 outside the window's rays gives the gradient of the window traced alone (same dL) - gradients within 2e-4 max."""
    dev = sg.medium.sigma_t.device
    dL = torch.zeros((batch.n_rays, 3), device=dev)
    dL[local_first:local_first + n] = torch.from_numpy(dLw).to(dev)
    samp = uivr.IndependentSampler(seed, spp)
    L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    grads = uivr.alloc_grads(sg)
    integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=dL, state_in=st, grads=grads)
    for k in (uivr.SIGMA_T_KEY, uivr.ALBEDO_KEY):
        ref = grads_window[k]
        tol = GRAD_RTOL * float(ref.abs().max()) + 1e-12
        err = float((grads[k] - ref).abs().max())
        assert float(ref.abs().max()) > 0 and err <= tol, f"masked full launch vs window, {k}: {err:.3e} > {tol:.3e}"


def _full_properties(uivr, sg, integ, spp, seed, shard=None):
    """Full-size launch: determinism, adjoint linearity, finite non-trivial gradients."""
    s = sg.sensors[0]
    n_pix = s.width * s.height
    shard = shard or uivr.ShardSpec()
    n_local = shard.n_local_pixels(n_pix)
    off, inter = shard.ray_mapping(spp)
    batch = uivr.RayBatch(n_rays=n_local * spp, spp=spp, sensor=s, ray_offset=off, interleave=inter)
    samp = uivr.IndependentSampler(seed, spp)
    L1, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    L2, _, _ = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    assert torch.equal(L1, L2)                                             # determinism
    assert torch.isfinite(L1).all() and float(L1.min()) >= 0.0
    gi = torch.randn((n_local, 3), device=L1.device) * 1e-6
    out = []
    for scale in (1.0, 2.0):
        L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)      # H1: primal then adjoint (path cache)
        dL = integ.film_backward(sg, (scale * gi).contiguous(), spp)
        grads = uivr.alloc_grads(sg)
        integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=dL, state_in=st, grads=grads)
        out.append(grads["_flat"])
    g1, g2 = out
    assert torch.isfinite(g1).all()
    scale = float(g1.abs().max())
    assert scale > 0
    assert float((g2 - 2.0 * g1).abs().max()) <= 1e-3 * scale             # linear in dL
    # energy bound: no emitter but the constant environment (radiance Le) and albedo <= 1 - the mean radiance cannot exceed it
    if isinstance(sg.emitter, uivr.ConstantEmitter):
        assert float(L1.double().mean()) <= 1.001 * float(max(sg.emitter.radiance))
    # every non-zero block of the gradient lies inside the support derived from sigma_t alone (the packing set of the multi-GPU all-reduce)
    from uivr_amd.distributed import COMPACT_BLOCK_FLOATS as B, _block_mask, gradient_support
    sup = gradient_support(sg.medium.sigma_t, grads)
    if sup is not None:
        n_full = (g1.numel() // B) * B
        assert int((_block_mask(g1[:n_full].view(-1, B)) & (1 - sup.mask)).sum()) == 0
    return L1, batch


# ------------------------------------------------------------------------------------------------------------
def test_config2_smoke_128_512x16(uivr, oracle, gpu):
    from uivr_amd import synthetic
    sg = synthetic.smoke_scene(res=128, film=512, device=gpu)
    props = props_for("drt")
    integ = _integrator(uivr, props)
    spp, seed = 16, 2002
    L, batch = _full_properties(uivr, sg, integ, spp, seed)
    assert L.shape[0] == 512 * 512 * 16
    # 1024 pixels of row 300, columns 192..: through the plume
    first = (300 * 512 + 192) * spp
    _window_check(uivr, oracle, sg, integ, props, spp, seed, L, first, first, 1024 * spp)
    # ... and a seeded window across the plume's silhouette: rays that miss the box, thin-edge rays, thicker ones
    n_pix = 256
    first = _seeded_window(sg, n_pix, 20021) * spp
    gw = _window_check(uivr, oracle, sg, integ, props, spp, seed, L, first, first, n_pix * spp, min_lookups_per_ray=0.0)
    _masked_full_equals_window(uivr, sg, integ, spp, seed, batch, first, n_pix * spp, gw, gw["_dL"], None)


def test_headline_dust_devil_256_512x32(uivr, oracle, gpu):
    from uivr_amd import synthetic
    sg = synthetic.dust_devil_scene(res=256, film=512, device=gpu)
    props = props_for("drt")
    integ = _integrator(uivr, props)
    spp, seed = 32, 2003
    L, batch = _full_properties(uivr, sg, integ, spp, seed)
    first = (380 * 512 + 200) * spp
    _window_check(uivr, oracle, sg, integ, props, spp, seed, L, first, first, 512 * spp)
    n_pix = 128
    first = _seeded_window(sg, n_pix, 20031) * spp
    gw = _window_check(uivr, oracle, sg, integ, props, spp, seed, L, first, first, n_pix * spp, min_lookups_per_ray=0.0)
    _masked_full_equals_window(uivr, sg, integ, spp, seed, batch, first, n_pix * spp, gw, gw["_dL"], None)


def test_headline_majorant_factor_8(uivr, oracle, gpu):
    """The reference's scenes run majorant_resolution_factor = 8 (scene_config.py:36): same checks on the supergrid path."""
    from uivr_amd import synthetic
    sg = synthetic.dust_devil_scene(res=256, film=512, device=gpu)
    sg.medium.majorant_resolution_factor = 8
    props = props_for("drt")
    integ = _integrator(uivr, props)
    spp, seed = 32, 2004
    L, batch = _full_properties(uivr, sg, integ, spp, seed)
    first = (380 * 512 + 200) * spp
    _window_check(uivr, oracle, sg, integ, props, spp, seed, L, first, first, 256 * spp, min_lookups_per_ray=0.5)
    n_pix = 128
    first = _seeded_window(sg, n_pix, 20041) * spp
    gw = _window_check(uivr, oracle, sg, integ, props, spp, seed, L, first, first, n_pix * spp, min_lookups_per_ray=0.0)
    _masked_full_equals_window(uivr, sg, integ, spp, seed, batch, first, n_pix * spp, gw, gw["_dL"], None)


def test_headline_quadratic_drt_factor_8(uivr, oracle, gpu):
    """`volpathsimple-drt-quadratic` (opt_config.py:123-169, the paper's comparison estimator) on the headline scene at the
    reference's majorant_resolution_factor 8, at 8 spp: the queued tracer's QUAD adjoint kernels (drt_sq.hip) - determinism,
    linearity of the adjoint, a window of rays against the oracle (radiance bit-exact, counters equal, gradients close)."""
    from uivr_amd import synthetic
    sg = synthetic.dust_devil_scene(res=256, film=512, device=gpu)
    sg.medium.majorant_resolution_factor = 8
    props = props_for("quadratic")
    integ = _integrator(uivr, props)
    spp, seed = 8, 2014
    L, _ = _full_properties(uivr, sg, integ, spp, seed)
    first = (380 * 512 + 200) * spp
    _window_check(uivr, oracle, sg, integ, props, spp, seed, L, first, first, 256 * spp, min_lookups_per_ray=0.5)


@pytest.mark.parametrize("factor", [0, 8])
def test_config4_512_rank0_share_of_1024x64(uivr, oracle, gpu, factor):
    """512^3 grid (16384 reduction tiles), rank 0 of 8 of a 1024^2 x 64 spp image: 8.4 M rays of the
    interleaved chunk map, RNG keyed by the global ray index.  With ONE majorant and at the reference's default
    majorant_resolution_factor 8: a 64^3 supergrid, whose majorants the queued tracer reads from L2 (drt_sq.hip, MG)."""
    from uivr_amd import synthetic
    sg = synthetic.dust_devil_scene(res=512, film=1024, device=gpu)
    sg.medium.majorant_resolution_factor = factor
    props = props_for("drt")
    integ = _integrator(uivr, props)
    spp, seed = 64, 2005
    shard = uivr.ShardSpec(0, 8, chunk_pixels=2048)
    L, batch = _full_properties(uivr, sg, integ, spp, seed, shard)
    assert L.shape[0] == 1024 * 1024 * 64 // 8
    # chunk 320 (2048 px = rows 640, 641) belongs to rank 0 (320 % 8 == 0) and is its local chunk 40
    chunk_rays = 2048 * spp
    px0 = 400                                                 # columns 400..527 of row 640
    local_first = 40 * chunk_rays + px0 * spp
    global_first = 320 * chunk_rays + px0 * spp
    n = 128 * spp
    # the window alone is traced UNSHARDED at the global offset: same rays, same streams
    _window_check(uivr, oracle, sg, integ, props, spp, seed, L, local_first, global_first, n)
    # a seeded window on the silhouette, inside one of rank 0's chunks (chunk c = pixels [2048 c, 2048 (c + 1)), c % 8 == 0)
    n_pix = 64
    p0 = _seeded_window(sg, n_pix, 20051 + factor, ranges=[(2048 * c, 2048 * (c + 1)) for c in range(0, 512, 8)])
    c, within = divmod(p0, 2048)
    assert c % 8 == 0
    local_first, global_first = (c // 8) * chunk_rays + within * spp, p0 * spp
    gw = _window_check(uivr, oracle, sg, integ, props, spp, seed, L, local_first, global_first, n_pix * spp, min_lookups_per_ray=0.0)
    _masked_full_equals_window(uivr, sg, integ, spp, seed, batch, local_first, n_pix * spp, gw, gw["_dL"], None)


def test_config3_optimize_loop_256_63_sensors(uivr, oracle, gpu):
    """dust devil 256^3, 63 sensors 512^2, batch 32768 px, spp_grad 16, spp_primal 1024 (reproduce.py:48-52):
    batch rays and radiance of a window bit-exact vs the oracle, window gradients, then the loop itself: 200 iterations
    against reference renderings of the target volume, loss decrease asserted."""
    from uivr_amd import synthetic
    sg = synthetic.dust_devil_scene(res=256, film=512, device=gpu, n_sensors=63)
    props = props_for("drt")
    integ = _integrator(uivr, props)
    B, spp, spp_grad, seed, seed_grad = 32768, 1024, 16, 3001, 3002
    params = {k: v.clone().requires_grad_(True) for k, v in sg.params().items() if k in integ.param_keys}
    image, _, _, sidx, pix = uivr.render_batch(B, sg, params=params, integrator=integ, seed=seed, seed_grad=seed_grad,
                                               spp=spp, spp_grad=spp_grad)
    assert image.shape == (B, 3) and torch.isfinite(image).all()
    assert len(torch.unique(sidx)) == 63
    # oracle: the first 24 batch entries at full spp (24576 rays), rays and radiance bit-exact
    W = 24
    sub0, sub1, sub2 = (uivr.sample_tea_32(seed, 17 * k + 5)[0] for k in (0, 1, 2))
    ro, rd, si_r, px_r = oracle.batch_sample_rays(sg.sensors, W, spp, sub0, sub1)
    np.testing.assert_array_equal(sidx[:W].cpu().numpy().astype(np.uint32), si_r)
    np.testing.assert_array_equal(pix[:W].cpu().numpy().astype(np.uint32), px_r)
    osc = oracle.OracleScene(_cpu_scene(uivr, sg), sensor_index=None)
    Lr, _ = oracle.render_primal(osc, props, spp, seed, rays_o=ro, rays_d=rd)
    np.testing.assert_allclose(image[:W].detach().cpu().numpy(), oracle.develop(Lr, spp), rtol=0, atol=1e-6)
    # gradients of a window of the adjoint rays: the first 2048 entries, spp_grad rays each
    ref = torch.full((63, 512, 512, 3), 0.5, device=gpu)
    ref_values = uivr.gather_ref_values(ref, sidx, pix)
    loss = uivr.losses.l1(image, ref_values)
    loss.backward()
    assert torch.isfinite(params[uivr.SIGMA_T_KEY].grad).all() and float(params[uivr.SIGMA_T_KEY].grad.abs().max()) > 0
    Wg = 2048
    ro2, rd2, _, _ = oracle.batch_sample_rays(sg.sensors, Wg, spp_grad, sub0, sub2)
    L2, _ = oracle.render_primal(osc, props, spp_grad, seed_grad, rays_o=ro2, rays_d=rd2)
    rng = np.random.default_rng(9)
    dLw = ((rng.random((Wg * spp_grad, 3), dtype=np.float32) - 0.5) * 1e-3).astype(np.float32)
    gs, ga, c_adj = oracle.render_backward(osc, props, spp_grad, seed_grad, dLw, L2, rays_o=ro2, rays_d=rd2)
    h = integ.native_handle(sg)
    table = uivr.sensors_to_device(sg.sensors, gpu)
    tro, trd, _, _ = uivr.sample_batch(integ, sg, table, Wg, spp_grad, seed, 2)
    np.testing.assert_array_equal(tro.cpu().numpy().view(np.uint32), ro2.view(np.uint32))
    batch = uivr.RayBatch(n_rays=Wg * spp_grad, spp=spp_grad, o=tro, d=trd)
    samp = uivr.IndependentSampler(seed_grad, spp_grad)
    Lw, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    np.testing.assert_array_equal(Lw.cpu().numpy().view(np.uint32), L2.view(np.uint32))
    h.enable_counters(True)
    h.reset_counters()
    grads = uivr.alloc_grads(sg)
    integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dLw).to(gpu), state_in=st, grads=grads)
    ca = {k: int(v) for k, v in h.get_counters().items()}
    h.enable_counters(False)
    assert ca == c_adj
    _close_on_device(grads[uivr.SIGMA_T_KEY], gs, "config 3 window grad sigma_t")
    _close_on_device(grads[uivr.ALBEDO_KEY], ga, "config 3 window grad albedo")
    del params, image, loss, grads

    # the optimisation loop as SURVEY.md 8d states it (reproduce.py:45-59; Adam, l1): reference renderings of the TARGET volume
    # (reduced ref_spp), 200 iterations from the constant init at the reference's default majorant_resolution_factor 8,
    # and the loss has to come down
    del ref, ref_values
    sc = uivr.SceneConfig(name="dust-devil", scene=sg, param_keys=[uivr.SIGMA_T_KEY, uivr.ALBEDO_KEY],
                          sensors=list(range(63)), start_from_value={uivr.SIGMA_T_KEY: 0.04, uivr.ALBEDO_KEY: 0.6},
                          majorant_resolution_factor=8, ref_spp=32)
    rendered = uivr.render_reference_image(sc, {s_: None for s_ in sc.sensors})
    ref = torch.stack([rendered[s_] for s_ in sc.sensors])
    assert ref.shape == (63, 512, 512, 3) and torch.isfinite(ref).all() and float(ref.std()) > 0
    oc = uivr.OptimizationConfig(name="t", spp=16, n_iter=200, lr=5e-3, primal_spp_factor=64, batch_size=32768)
    _, p, _, hist = uivr.run_optimization(None, oc, sc, "volpathsimple-drt", ref_images=ref)
    assert len(hist) == 200 and all(np.isfinite(hist))
    first, last = float(np.mean(hist[:20])), float(np.mean(hist[-20:]))
    assert last < 0.95 * first, (first, last)                                   # (measured: 0.1420 -> 0.1247)
    assert float(p[uivr.SIGMA_T_KEY].min()) >= 0 and float(p[uivr.SIGMA_T_KEY].max()) <= 250
    assert float(p[uivr.ALBEDO_KEY].min()) >= 0 and float(p[uivr.ALBEDO_KEY].max()) <= 1
    assert float((p[uivr.SIGMA_T_KEY] - 0.04).abs().max()) > 0                  # the parameters moved


def test_config3_as_reproduce_every_resolution_level(uivr, oracle, gpu):
    """The dust-devil DRT run as python/reproduce.py sets it up (:48-59, :108-110; bench.py config3_as_reproduce): the parameters start on a
    16^3 grid and are upsampled x2 four times (optimize.py:134-166, 228-252), the majorant supergrid follows (adjust_majorant_res_factor,
    optimize.py:182-199: factor 4 on 16^3, 8 from 32^3 on), the scene is lit by a 4096 x 2048 environment map (scene_config.py:152).  At EVERY
    level - a thin medium on that level's grid, as the optimisation sees it early on -: one batched iteration's rays and radiance bit-exact
    against the oracle for a window of batch entries, event counters equal, window gradients close; then the loop itself through all five
    levels (25 iterations, the reference's learning rate, schedule and upsampling fractions)."""
    from uivr_amd import synthetic
    target = synthetic.dust_devil_scene(res=256, film=512, device=gpu, n_sensors=63)
    g = torch.Generator().manual_seed(5)
    env = (torch.rand(2048, 4096, 3, generator=g) ** 4 * 3.0 + 0.2).to(gpu)
    target.emitter = uivr.EnvmapEmitter(pixels=env, scale=1.0)
    props = props_for("drt")
    integ = _integrator(uivr, props)
    table = uivr.sensors_to_device(target.sensors, gpu)
    B, spp, spp_grad, seed, seed_grad = 4096, 1024, 16, 3101, 3102
    full = target.medium.sigma_t
    for res in (16, 32, 64, 128, 256):
        # the level's grid: the target downsampled (max over the block: the structure survives) and thinned, albedo at its initial 0.6
        k = 256 // res
        st = full.reshape(res, k, res, k, res, k, 1).amax(dim=(1, 3, 5)) * 0.05 + 0.04 / 100
        factor = uivr.adjusted_majorant_res_factor(8, st.shape)
        assert factor == (4 if res == 16 else 8)
        medium = uivr.GridMedium(sigma_t=st.contiguous(), albedo=torch.full((res, res, res, 3), 0.6, device=gpu), bbox_min=target.medium.bbox_min,
                                 bbox_max=target.medium.bbox_max, scale=target.medium.scale, majorant_resolution_factor=factor)
        sg = uivr.Scene(medium=medium, emitter=target.emitter, sensors=target.sensors)
        params = {kk: v.clone().requires_grad_(True) for kk, v in sg.params().items() if kk in integ.param_keys}
        image, _, _, sidx, pix = uivr.render_batch(B, sg, params=params, integrator=integ, seed=seed + res, seed_grad=seed_grad + res,
                                                   spp=spp, spp_grad=spp_grad, sensor_table=table)
        assert image.shape == (B, 3) and torch.isfinite(image).all()
        W = 6
        sub0, sub1, sub2 = (uivr.sample_tea_32(seed + res, 17 * kq + 5)[0] for kq in (0, 1, 2))
        ro, rd, si_r, px_r = oracle.batch_sample_rays(sg.sensors, W, spp, sub0, sub1)
        np.testing.assert_array_equal(sidx[:W].cpu().numpy().astype(np.uint32), si_r)
        np.testing.assert_array_equal(pix[:W].cpu().numpy().astype(np.uint32), px_r)
        osc = oracle.OracleScene(_cpu_scene(uivr, sg), sensor_index=None)
        Lr, _ = oracle.render_primal(osc, props, spp, seed + res, rays_o=ro, rays_d=rd)
        np.testing.assert_allclose(image[:W].detach().cpu().numpy(), oracle.develop(Lr, spp), rtol=0, atol=1e-6)
        Wg = 192
        ro2, rd2, _, _ = oracle.batch_sample_rays(sg.sensors, Wg, spp_grad, sub0, sub2)
        L2, c_p = oracle.render_primal(osc, props, spp_grad, seed_grad + res, rays_o=ro2, rays_d=rd2)
        rng = np.random.default_rng(res)
        dLw = ((rng.random((Wg * spp_grad, 3), dtype=np.float32) - 0.5) * 1e-3).astype(np.float32)
        gs, ga, c_adj = oracle.render_backward(osc, props, spp_grad, seed_grad + res, dLw, L2, rays_o=ro2, rays_d=rd2)
        tro, trd, _, _ = uivr.sample_batch(integ, sg, table, Wg, spp_grad, seed + res, 2)
        np.testing.assert_array_equal(tro.cpu().numpy().view(np.uint32), ro2.view(np.uint32))
        batch = uivr.RayBatch(n_rays=Wg * spp_grad, spp=spp_grad, o=tro, d=trd)
        samp = uivr.IndependentSampler(seed_grad + res, spp_grad)
        h = integ.native_handle(sg)
        h.enable_counters(True)
        h.reset_counters()
        Lw, _, stt = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
        cp = {kk: int(v) for kk, v in h.get_counters().items()}
        np.testing.assert_array_equal(Lw.cpu().numpy().view(np.uint32), L2.view(np.uint32))
        assert cp == c_p
        h.reset_counters()
        grads = uivr.alloc_grads(sg)
        integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=torch.from_numpy(dLw).to(gpu), state_in=stt, grads=grads)
        ca = {kk: int(v) for kk, v in h.get_counters().items()}
        h.enable_counters(False)
        assert ca == c_adj
        _close_on_device(grads[uivr.SIGMA_T_KEY], gs, f"level {res}^3 window grad sigma_t")
        _close_on_device(grads[uivr.ALBEDO_KEY], ga, f"level {res}^3 window grad albedo")
        del params, image, grads, osc
    # the loop through all five levels as reproduce.py configures it
    target.medium.majorant_resolution_factor = 8
    sc = uivr.SceneConfig(name="dust-devil", scene=target, param_keys=[uivr.SIGMA_T_KEY, uivr.ALBEDO_KEY], sensors=list(range(63)),
                          start_from_value={uivr.SIGMA_T_KEY: 0.04 / 100, uivr.ALBEDO_KEY: 0.6}, majorant_resolution_factor=8, ref_spp=16)
    rendered = uivr.render_reference_image(sc, {s_: None for s_ in sc.sensors})
    ref = torch.stack([rendered[s_] for s_ in sc.sensors])
    oc = uivr.OptimizationConfig(name="r", spp=16, n_iter=25, lr=3e-4, primal_spp_factor=64, batch_size=32768,
                                 lr_schedule=uivr.Schedule.Last25, upsample=[0.04, 0.16, 0.36, 0.64])
    assert sorted(oc.upsample_at) == [1, 4, 9, 16]
    _, p, _, hist = uivr.run_optimization(None, oc, sc, "volpathsimple-drt", ref_images=ref)
    assert len(hist) == 25 and all(np.isfinite(hist))
    assert tuple(p[uivr.SIGMA_T_KEY].shape) == (256, 256, 256, 1) and tuple(p[uivr.ALBEDO_KEY].shape) == (256, 256, 256, 3)
    assert float(p[uivr.SIGMA_T_KEY].min()) >= 0 and float(p[uivr.ALBEDO_KEY].min()) >= 0 and float(p[uivr.ALBEDO_KEY].max()) <= 1
    assert float((p[uivr.SIGMA_T_KEY] - 0.04 / 100).abs().max()) > 0


def test_config5_nerf_256_512x32_128_queries(uivr, oracle, gpu):
    """The registered `nerf` IntegratorConfig (128 queries, opt_config.py:162-169) on the 256^3 density + emission
    grids (emission = the albedo grid, scene_config.py:109-110) at 512^2 x 32 spp."""
    from uivr_amd import synthetic
    sg = synthetic.dust_devil_scene(res=256, film=512, device=gpu)
    sg.medium.emission = sg.medium.albedo
    integ = uivr.get_int_config("nerf").create(max_depth=64)
    assert integ.queries_per_ray == 128
    spp, seed = 32, 2006
    s = sg.sensors[0]
    batch = uivr.RayBatch(n_rays=512 * 512 * spp, spp=spp, sensor=s)
    samp = uivr.IndependentSampler(seed, spp)
    L, _, st = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    L2, _, _ = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), batch)
    assert torch.equal(L, L2) and torch.isfinite(L).all()
    gi = torch.randn((512 * 512, 3), device=gpu) * 1e-6
    g = []
    for scale in (1.0, 2.0):
        grads = uivr.alloc_grads(sg, integ.param_keys)
        integ.sample(uivr.ADMode.Backward, sg, samp, batch, δL=integ.film_backward(sg, (scale * gi).contiguous(), spp),
                     state_in=st, grads=grads)
        g.append(grads["_flat"])
    sc = float(g[0].abs().max())
    assert sc > 0 and float((g[1] - 2.0 * g[0]).abs().max()) <= 1e-3 * sc
    del g, grads
    # window vs oracle: 256 pixels of row 380
    first, n = (380 * 512 + 200) * spp, 256 * spp
    cpu = _cpu_scene(uivr, sg)
    osc = oracle.OracleScene(cpu)
    em = cpu.medium.emission
    props = dict(queries_per_ray=128)
    Lr, c_p = oracle.nerf_render(osc, em, props, spp, seed, n_rays=n, ray_offset=first)
    np.testing.assert_array_equal(L[first:first + n].cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    rng = np.random.default_rng(6)
    dL = ((rng.random((n, 3), dtype=np.float32) - 0.5) * 1e-3).astype(np.float32)
    gs, ge, c_a = oracle.nerf_render(osc, em, props, spp, seed, dL=dL, L_in=Lr, n_rays=n, ray_offset=first)
    wb = uivr.RayBatch(n_rays=n, spp=spp, sensor=s, ray_offset=first)
    h = integ.native_handle(sg)
    Lw, _, stw = integ.sample(uivr.ADMode.Primal, sg, samp.clone(), wb)
    np.testing.assert_array_equal(Lw.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    h.enable_counters(True)
    h.reset_counters()
    grads = uivr.alloc_grads(sg, integ.param_keys)
    integ.sample(uivr.ADMode.Backward, sg, samp, wb, δL=torch.from_numpy(dL).to(gpu), state_in=stw, grads=grads)
    ca = {k: int(v) for k, v in h.get_counters().items()}
    h.enable_counters(False)
    assert ca == c_a
    _close_on_device(grads[uivr.SIGMA_T_KEY], gs, "nerf window grad sigma_t")
    _close_on_device(grads[uivr.EMISSION_KEY], ge, "nerf window grad emission")


def test_supergrid_tracer_repeats_bitwise_at_size(gpu):
    """tools/stress_super.py: the supergrid tracer's flights travel through LDS slots and masks shared by the twelve waves
    of a workgroup - a lost, duplicated or stale flight would change rays.  The headline scene at factor 8 (and three
    smaller shapes that end their kernels differently), the same seeds over and over: radiance bitwise the same every time,
    gradients equal up to summation order and finite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_super.py"), "--reps", "9"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "STRESS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
