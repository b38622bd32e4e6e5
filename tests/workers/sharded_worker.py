"""Worker of tests/test_gpu_sharded.py: one of `world` processes (torch.distributed.run), all on cuda:0 with the
gloo backend - the PRODUCT path (HIP kernels through the C ABI) under a real process group.

Checks, on every rank:
  * sharded `render_batch` (batch entries dealt across ranks, random streams keyed by the global entry / ray
    index): the union of the ranks' images equals the unsharded image BITWISE, sensor / pixel indices equal;
  * the all-reduced gradient grids equal the unsharded gradients (fp32 summation order only);
  * an unsharded call under the initialised process group does NOT communicate (gradients not multiplied by world);
  * sharded sensor-centric `render` likewise;
  * `run_optimization(shard=...)`: same parameters after 4 iterations as the unsharded loop, identical on all ranks.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import uivr_amd as u
from uivr_amd import synthetic

GRAD_RTOL = 2e-4


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    scene = synthetic.smoke_scene(res=24, film=32, device=dev, optical_side=10.0)
    scene.medium.majorant_resolution_factor = int(os.environ.get("DRT_TEST_FACTOR", "0"))   # > 0: the supergrid tracer (drt_super.hip)
    scene.sensors = synthetic.ring_sensors(5, radius=5.0, height=0.8, fov=30.0, width=32, film_height=32)
    integ = u.get_int_config("volpathsimple-drt").create(max_depth=32)
    shard = u.ShardSpec(rank, world)
    B, spp, spp_grad, seed, seed_grad = 1001, 4, 2, 11, 12            # ragged: 1001 entries over 2 ranks

    def run(sh):
        params = {k: v.clone().requires_grad_(True) for k, v in scene.params().items() if k in integ.param_keys}
        image, _, _, sidx, pix = u.render_batch(B, scene, params=params, integrator=integ, seed=seed, seed_grad=seed_grad,
                                                spp=spp, spp_grad=spp_grad, shard=sh)
        ref = torch.full_like(image, 0.3)
        n_local = image.shape[0]
        loss = u.losses.l2(image, ref) * u.local_loss_scale(n_local, B)
        loss.backward()
        return image.detach(), sidx, pix, {k: p.grad for k, p in params.items()}, loss.detach()

    img_u, sidx_u, pix_u, g_u, loss_u = run(None)                      # unsharded, under the process group
    img_s, sidx_s, pix_s, g_s, loss_s = run(shard)
    first, count = shard.batch_range(B)
    assert img_s.shape[0] == count
    assert torch.equal(img_s, img_u[first:first + count]), "sharded image differs from its slice of the unsharded one"
    assert torch.equal(sidx_s, sidx_u[first:first + count]) and torch.equal(pix_s, pix_u[first:first + count])
    total = u.allreduce_scalar(loss_s)
    assert abs(float(total) - float(loss_u)) <= 1e-6 * abs(float(loss_u)) + 1e-12
    for k in g_u:
        tol = GRAD_RTOL * float(g_u[k].abs().max()) + 1e-12
        err = float((g_s[k] - g_u[k]).abs().max())
        assert err <= tol, (k, err, tol)
    # unsharded twice gives the same gradients: no hidden all-reduce multiplied them by `world`
    _, _, _, g_u2, _ = run(None)
    for k in g_u:
        assert float((g_u2[k] - g_u[k]).abs().max()) <= GRAD_RTOL * float(g_u[k].abs().max())

    # sensor-centric render, sharded
    n_pix = 32 * 32
    sh = u.ShardSpec(rank, world, u.ShardSpec.default_chunk(n_pix, world, 64))
    def run_render(s):
        params = {k: v.clone().requires_grad_(True) for k, v in scene.params().items() if k in integ.param_keys}
        img = u.render(scene, params=params, integrator=integ, sensor=1, spp=4, seed=5, seed_grad=6, shard=s)
        n_loc = img.shape[0]
        (((img - 0.4) ** 2).sum() / (n_pix * 3)).backward()           # global normalisation
        return img.detach(), {k: p.grad for k, p in params.items()}
    img_u, g_u = run_render(None)
    img_s, g_s = run_render(sh)
    assert torch.equal(img_s, img_u[sh.pixel_indices(n_pix, dev)])
    for k in g_u:
        assert float((g_s[k] - g_u[k]).abs().max()) <= GRAD_RTOL * float(g_u[k].abs().max()) + 1e-12

    # the compacted all-reduce on device buffers (native block mask): same sums as the dense collective
    from uivr_amd import distributed as D
    Bf = D.COMPACT_BLOCK_FLOATS
    gen = torch.Generator().manual_seed(99 + rank)
    host = torch.randn(300, Bf, generator=gen) * (torch.rand(300, 1, generator=gen) < 0.15)
    flat = torch.cat([host.reshape(-1), torch.ones(21)]).to(dev)
    if rank == world - 1:
        flat[11 * Bf + 5] = float("inf")
    want = flat.clone()
    dist.all_reduce(want)
    for mode, expect in (("auto", "compact"), ("always", "compact"), ("never", "dense")):
        f, st = flat.clone(), {}
        u.allreduce_gradients({"_flat": f}, compact=mode, stats=st)
        assert torch.equal(f, want), mode
        assert st["mode"] == expect and (expect == "dense" or st["floats"] < 0.4 * flat.numel()), (mode, st)
    # ... and on the gradient of a render: sparse volume (a small blob in an empty grid) -> compact mode, same result
    blob = synthetic.smoke_scene(res=24, film=32, device=dev, optical_side=10.0)
    sig = torch.zeros_like(blob.medium.sigma_t)
    sig[8:14, 8:14, 8:14] = blob.medium.sigma_t[8:14, 8:14, 8:14] + 0.5
    blob.medium.sigma_t = sig
    def blob_grads(mode):
        g = u.alloc_grads(blob)
        n_px = 32 * 32
        s2 = u.ShardSpec(rank, world, u.ShardSpec.default_chunk(n_px, world, 64))
        off, inter = s2.ray_mapping(4)
        batch = u.RayBatch(n_rays=s2.n_local_pixels(n_px) * 4, spp=4, sensor=blob.sensors[0], ray_offset=off, interleave=inter)
        smp = u.IndependentSampler(7, 4)
        L, _, state = integ.sample(u.ADMode.Primal, blob, smp.clone(), batch)
        integ.sample(u.ADMode.Backward, blob, smp, batch, δL=torch.full_like(L, 1e-3), state_in=state, grads=g)
        st = {}
        u.allreduce_gradients(g, compact=mode, stats=st)
        return g["_flat"], st
    g_dense, st_d = blob_grads("never")
    g_comp, st_c = blob_grads("auto")
    assert st_d["mode"] == "dense"
    if rank == 0:
        print("blob gradient all-reduce:", st_c, flush=True)
    assert float((g_comp - g_dense).abs().max()) <= GRAD_RTOL * float(g_dense.abs().max())

    # the optimisation loop
    sc = u.SceneConfig(name="s", scene=scene, param_keys=[u.SIGMA_T_KEY, u.ALBEDO_KEY], sensors=list(range(5)),
                       start_from_value={u.SIGMA_T_KEY: 0.4, u.ALBEDO_KEY: 0.6}, max_depth=16, ref_spp=64, max_density=20.0)
    oc = u.OptimizationConfig("t", spp=2, n_iter=4, lr=2e-2, primal_spp_factor=2, batch_size=512)
    ref = torch.full((5, 32, 32, 3), 0.5, device=dev)
    _, p_u, _, h_u = u.run_optimization(None, oc, sc, "volpathsimple-drt", ref_images=ref)
    _, p_u2, _, h_u2 = u.run_optimization(None, oc, sc, "volpathsimple-drt", ref_images=ref)
    np.testing.assert_allclose(h_u2, h_u, rtol=1e-4)           # (gradient sums differ in the last bits from run to run)
    # first iteration of the loop by hand, sharded vs unsharded: constant initial grids, supergrid, l1 loss
    from uivr_amd.optimize import _scene_with, adjusted_majorant_res_factor
    p0 = {u.SIGMA_T_KEY: torch.full((24, 24, 24, 1), 0.4, device=dev), u.ALBEDO_KEY: torch.full((24, 24, 24, 3), 0.6, device=dev)}
    sc0 = _scene_with(u.Scene(scene.medium, scene.emitter, scene.sensors), p0, adjusted_majorant_res_factor(8, (24, 24, 24, 1)))
    integ16 = u.get_int_config("volpathsimple-drt").create(max_depth=16)
    def first_iteration(sh):
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p0.items()}
        image, _, _, si, pi = u.render_batch(512, sc0, params=leaves, integrator=integ16, spp=4, spp_grad=2,
                                             seed=u.sample_tea_32(0, 988378)[0], seed_grad=u.sample_tea_32(1, 988378)[0], shard=sh)
        rv = u.gather_ref_values(ref, si, pi)
        (u.losses.l1(image, rv) * u.local_loss_scale(image.shape[0], 512)).backward()
        return torch.cat([leaves[k].grad.reshape(-1) for k in leaves])
    gu, gs_ = first_iteration(None), first_iteration(shard)
    rel = float((gu - gs_).abs().max() / gu.abs().max())
    if rank == 0:
        print("first-iteration gradient: max rel diff sharded vs unsharded", rel, "max|g|", float(gu.abs().max()), flush=True)
    assert rel < 1e-3, rel
    _, p_s, _, h_s = u.run_optimization(None, oc, sc, "volpathsimple-drt", ref_images=ref, shard=shard)
    if rank == 0:
        print("histories", h_u, h_s, flush=True)
    np.testing.assert_allclose(h_s, h_u, rtol=1e-4)
    for k in p_u:
        # Adam normalises the step: a gradient that differs in the last bits moves a parameter by (almost) the same lr
        d = float((p_s[k] - p_u[k]).abs().max())
        assert d <= 2e-3 * 2e-2 * 4 + 1e-7, (k, d)
        other = p_s[k].clone()
        dist.broadcast(other, src=0)
        assert torch.equal(other, p_s[k]), "ranks diverged"
    dist.barrier()
    if rank == 0:
        print("SHARDED_WORKER_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
