"""Device primitives vs the oracle, bit for bit (drt_debug_eval hook of the C ABI).
These are the [M3-ext] building blocks restated in DESIGN.md: PCG32/TEA, the fixed
log / sincos polynomials, sphere warping, trilinear grid lookup, box intersection,
sensor rays, and IEEE division / sqrt."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import props_for

pytestmark = pytest.mark.gpu


def _eval(uivr, gpu, scene, op, inp):
    sg = uivr.scene_to(scene, gpu)
    integ = uivr.load_dict(dict(type="volpathsimple", **props_for("drt")))
    h = integ.native_handle(sg)
    if scene.sensors:
        integ._set_rays(h, uivr.RayBatch(n_rays=1, spp=1, sensor=scene.sensors[0]))
    n = inp.shape[0]
    buf = np.zeros((n, 6), dtype=np.float32)
    buf[:, :inp.shape[1]] = inp
    tin = torch.from_numpy(buf).to(gpu)
    tout = torch.empty_like(tin)
    h.debug_eval(op, tin.data_ptr(), n, tout.data_ptr())
    torch.cuda.synchronize()
    return tout.cpu().numpy()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_log_sincos_sphere(uivr, oracle, gpu):
    L = oracle.lib()
    scene = uivr.cube_test_scene(8, 8)
    k = np.arange(1, 1 << 23, 997, dtype=np.int64)
    x = (k.astype(np.float64) / float(1 << 23)).astype(np.float32)       # the values 1-u can take
    x = np.concatenate([x, np.float32([1.0, 2.0 ** -23, 0.5, 0.70710677, 0.7071068])])
    got = _eval(uivr, gpu, scene, 0, x[:, None])[:, 0]
    ref = np.array([L.drto_logf(float(v)) for v in x], dtype=np.float32)
    np.testing.assert_array_equal(_bits(got), _bits(ref))

    u = np.concatenate([np.random.default_rng(1).random(20000, dtype=np.float32),
                        np.float32([0, 0.25, 0.5, 0.75, 0.125, 0.375, 1 - 2.0 ** -23])])
    got = _eval(uivr, gpu, scene, 1, u[:, None])[:, :2]
    s, c = C.c_float(), C.c_float()
    ref = np.zeros_like(got)
    for i, v in enumerate(u):
        L.drto_sincos_2pi(float(v), C.byref(s), C.byref(c))
        ref[i] = (s.value, c.value)
    np.testing.assert_array_equal(_bits(got), _bits(ref))

    uv = np.random.default_rng(2).random((20000, 2), dtype=np.float32)
    got = _eval(uivr, gpu, scene, 2, uv)[:, :3]
    out = (C.c_float * 3)()
    ref = np.zeros_like(got)
    for i, (a, b) in enumerate(uv):
        L.drto_uniform_sphere(float(a), float(b), out)
        ref[i] = out[:]
    np.testing.assert_array_equal(_bits(got), _bits(ref))


def test_pcg32_and_sensor(uivr, oracle, gpu):
    L = oracle.lib()
    scene = uivr.cube_test_scene(128, 128)
    rng = np.random.default_rng(3)
    seeds = rng.integers(0, 2 ** 32, size=(2000, 2), dtype=np.uint64).astype(np.uint32)
    got = _eval(uivr, gpu, scene, 6, seeds.view(np.float32))
    ref = np.zeros((2000, 6), dtype=np.float32)
    for i, (s, idx) in enumerate(seeds):
        L.drto_pcg32_floats(int(s), int(idx), 6, ref[i].ctypes.data_as(C.POINTER(C.c_float)))
    np.testing.assert_array_equal(_bits(got), _bits(ref))

    osc = oracle.OracleScene(scene)
    pix = rng.integers(0, 128 * 128, size=5000, dtype=np.uint64).astype(np.uint32)
    uv = rng.random((5000, 2), dtype=np.float32)
    inp = np.concatenate([pix.view(np.float32)[:, None], uv], axis=1)
    got = _eval(uivr, gpu, scene, 7, inp)
    ref = np.zeros((5000, 6), dtype=np.float32)
    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    for i in range(5000):
        L.drto_sensor_ray(C.byref(osc.sensor), int(pix[i]), float(uv[i, 0]), float(uv[i, 1]), o, d)
        ref[i, :3] = o[:]
        ref[i, 3:] = d[:]
    np.testing.assert_array_equal(_bits(got), _bits(ref))


def test_grid_lookup_and_box(uivr, oracle, gpu):
    L = oracle.lib()
    rng = np.random.default_rng(4)
    res = (5, 7, 6)   # X, Y, Z - ragged on purpose
    sigma_t = rng.random((res[2], res[1], res[0], 1), dtype=np.float32) * 3
    albedo = rng.random((res[2], res[1], res[0], 3), dtype=np.float32)
    medium = uivr.GridMedium(sigma_t=sigma_t, albedo=albedo, bbox_min=(-1, 0, 2), bbox_max=(1.5, 3, 2.5), scale=1.7)
    scene = uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter(), sensors=[])
    osc = oracle.OracleScene(scene, sensor_index=None)
    lo, hi = np.float32(medium.bbox_min), np.float32(medium.bbox_max)
    p = (lo + (hi - lo) * (rng.random((20000, 3), dtype=np.float32) * 1.1 - 0.05)).astype(np.float32)  # incl. slightly outside
    got = _eval(uivr, gpu, scene, 3, p)[:, 0]
    ref = np.array([L.drto_eval_sigma_t(C.byref(osc.medium), (C.c_float * 3)(*q)) for q in p], dtype=np.float32)
    np.testing.assert_array_equal(_bits(got), _bits(ref))
    got = _eval(uivr, gpu, scene, 4, p)[:, :3]
    out = (C.c_float * 3)()
    ref = np.zeros_like(got)
    for i, q in enumerate(p):
        L.drto_eval_albedo(C.byref(osc.medium), (C.c_float * 3)(*q), out)
        ref[i] = out[:]
    np.testing.assert_array_equal(_bits(got), _bits(ref))

    o = (rng.normal(size=(20000, 3)) * 2 + (lo + hi) / 2).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50, 0] = 0.0   # axis-parallel rays
    got = _eval(uivr, gpu, scene, 5, np.concatenate([o, d], axis=1))
    ref = np.zeros((20000, 5), dtype=np.float32)
    t, nrm = C.c_float(), (C.c_float * 3)()
    for i in range(20000):
        v = L.drto_box_hit(C.byref(osc.medium), (C.c_float * 3)(*o[i]), (C.c_float * 3)(*d[i]), C.byref(t), nrm)
        ref[i] = (float(v), t.value, nrm[0], nrm[1], nrm[2])
    np.testing.assert_array_equal(_bits(got[:, :5]), _bits(ref))


def test_ieee_div_sqrt(uivr, gpu):
    scene = uivr.cube_test_scene(8, 8)
    rng = np.random.default_rng(5)
    a = (rng.random((50000, 3), dtype=np.float32) * 10 + 1e-3).astype(np.float32)
    got = _eval(uivr, gpu, scene, 8, a)
    np.testing.assert_array_equal(_bits(got[:, 1]), _bits(a[:, 0] / a[:, 1]))
    np.testing.assert_array_equal(_bits(got[:, 2]), _bits(np.sqrt(a[:, 0])))
    a2, b2 = a[:, 0] * a[:, 0], a[:, 1] * a[:, 1]
    np.testing.assert_array_equal(_bits(got[:, 0]), _bits(a2 / (a2 + b2)))


@pytest.mark.parametrize("factor", [0, 4])
def test_e2_sample_interaction_drt_matches_oracle(uivr, oracle, gpu, factor):
    """E2 (Medium::sample_interaction_drt, volpathsimple.py:549-551) on the device == the oracle's restatement, walk
    for walk (debug op 14 / drto_sample_interaction_drt: same ray, stream PCG32(tea32(0x5eed, i))): valid flag, selected
    distance t', weight W and maxt bit for bit.  The restatement itself is pinned by tests/test_oracle_e2.py."""
    rng = np.random.default_rng(5)
    st = (rng.random((16, 16, 16, 1), dtype=np.float32) ** 3 * 6.0).astype(np.float32)
    st[:, :, 5:9] = 0.0
    medium = uivr.GridMedium(sigma_t=st, albedo=np.full((16, 16, 16, 3), 0.5, np.float32), bbox_min=(-1, -1, -1),
                             bbox_max=(1, 1, 1), scale=1.2, majorant_resolution_factor=factor)
    scene = uivr.Scene(medium=medium, emitter=uivr.ConstantEmitter(), sensors=[])
    osc = oracle.OracleScene(scene, sensor_index=None)
    n = 4096
    o = (rng.random((n, 3), dtype=np.float32) * 1.8 - 0.9).astype(np.float32)          # inside the box
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    got = _eval(uivr, gpu, scene, 14, np.concatenate([o, d], axis=1))
    n_valid = 0
    for i in range(n):
        valid, t, W, maxt = oracle.sample_interaction_drt(osc, o[i], d[i], 0x5eed, 1, first=i)
        assert bool(got[i, 0]) == bool(valid[0]), i
        np.testing.assert_array_equal(_bits(got[i, 1:4]), _bits(np.float32([t[0], W[0], maxt])), err_msg=str(i))
        n_valid += int(valid[0])
    assert n_valid > n // 2


@pytest.mark.parametrize("block", [64, 128, 256])
def test_grad_block_mask_matches_definition(uivr, gpu, block):
    """drt_grad_block_mask (the compacted gradient all-reduce, distributed.py): 1 for every block that holds anything
    but zeros - NaN, inf, denormals and a lone last element included; -0.0 is zero."""
    native = uivr._native.native          # (attribute access: importing `uivr_amd._native` by name would load it twice)
    gen = torch.Generator().manual_seed(block)
    for n_blocks in (1, 3, 257, 4099):
        body = torch.randn(n_blocks, block, generator=gen) * (torch.rand(n_blocks, 1, generator=gen) < 0.3)
        if n_blocks > 3:
            body[1] = 0; body[1, block - 1] = 1e-42          # denormal in the last lane
            body[2] = 0; body[2, 0] = -0.0
            body[5] = 0; body[5, 17] = float("nan")
            body[6] = 0; body[6, block // 2] = float("-inf")
        want = (body != 0).any(dim=1).to(torch.uint8)
        dev_body = body.to(gpu)
        mask = torch.full((n_blocks + 8,), 7, dtype=torch.uint8, device=gpu)     # guard bytes after the mask
        native().grad_block_mask(torch.cuda.current_stream().cuda_stream, dev_body.data_ptr(), n_blocks, block, mask.data_ptr())
        assert torch.equal(mask[:n_blocks].cpu(), want)
        assert bool((mask[n_blocks:] == 7).all())
    with pytest.raises(RuntimeError):
        native().grad_block_mask(0, dev_body.data_ptr(), 1, 96, mask.data_ptr())


@pytest.mark.parametrize("variant,factor", [("drt", 0), ("drt", 4), ("basic", 0), ("quadratic", 0)])
def test_gradient_support_covers_every_nonzero_gradient_block(uivr, gpu, variant, factor):
    """distributed.gradient_support (the packing set of the ONE-collective gradient all-reduce, known before the adjoint
    pass from sigma_t alone): every non-zero 256-byte block of the gradient buffer the adjoint pass produces lies inside
    it - for the registered estimators, with and without a majorant supergrid - and it is a real restriction (the albedo
    planes of a sparse volume are mostly outside)."""
    from conftest import props_for
    from uivr_amd import synthetic
    from uivr_amd.distributed import COMPACT_BLOCK_FLOATS as B, _block_mask, gradient_support
    sg = synthetic.smoke_scene(res=48, film=64, device=gpu, optical_side=10.0)
    sg.medium.majorant_resolution_factor = factor
    assert float((sg.medium.sigma_t == 0).float().mean()) > 0.3                 # a sparse volume
    integ = uivr.load_dict(dict(type="volpathsimple", **props_for(variant)))
    spp = 8
    gi = (torch.rand((64 * 64, 3), device=gpu) - 0.5) * 1e-2
    for seed in (3, 4):
        grads = uivr.render_backward(sg, integ, gi, sensor=0, spp=spp, seed=seed)
        sup = gradient_support(sg.medium.sigma_t, grads, sparse_keys=(uivr.ALBEDO_KEY,))
        assert sup is not None
        flat = grads["_flat"]
        n_full = (flat.numel() // B) * B
        got = _block_mask(flat[:n_full].view(-1, B))
        assert int((got & (1 - sup.mask)).sum()) == 0
        assert float(grads[uivr.ALBEDO_KEY].abs().max()) > 0 and float(grads[uivr.SIGMA_T_KEY].abs().max()) > 0
        assert sup.count < 0.8 * sup.mask.numel()


@pytest.mark.parametrize("shape", [(48, 48, 48), (50, 29, 37), (5, 70, 9), (33, 2, 1)])
def test_device_gradient_support_equals_the_torch_formulation(uivr, gpu, shape):
    """drt_grad_support_mask (two small kernels) gives exactly the mask of the torch formulation of distributed.gradient_support
    (3 x 3 x 3 dilation of sigma_t != 0 over the albedo plane, every block of the sigma_t plane and of the padding) - grids whose
    rows are not multiples of 32 voxels, blocks that span several rows, non-zero voxels on the borders."""
    from uivr_amd.distributed import gradient_support
    rz, ry, rx = shape
    gen = torch.Generator().manual_seed(rx * 131 + ry)
    st = torch.zeros(rz, ry, rx, 1)
    idx = torch.randint(0, rz * ry * rx, (max(2, rz * ry * rx // 400),), generator=gen)
    st.view(-1)[idx] = torch.rand(idx.numel(), generator=gen) + 0.1
    st[0, 0, 0, 0] = 1.0; st[rz - 1, ry - 1, rx - 1, 0] = 1.0
    scene = uivr.cube_test_scene(8, 8)
    scene.medium.sigma_t, scene.medium.albedo = st, torch.rand(rz, ry, rx, 3, generator=gen)
    cpu_grads = uivr.alloc_grads(scene)
    ref = gradient_support(st, cpu_grads, sparse_keys=(uivr.ALBEDO_KEY,))
    sg = uivr.scene_to(scene, gpu)
    dev_grads = uivr.alloc_grads(sg)
    got = gradient_support(sg.medium.sigma_t, dev_grads, sparse_keys=(uivr.ALBEDO_KEY,))
    if ref is None:
        assert got is None
        return
    assert got.mask.is_cuda and torch.equal(got.mask.cpu(), ref.mask) and got.count == ref.count


@pytest.mark.parametrize("block", [64, 128, 256])
def test_grad_pack_kernels_equal_the_torch_formulation(uivr, gpu, block):
    """drt_grad_block_positions / drt_grad_pack / drt_grad_unpack (the packing of the one-collective gradient all-reduce):
    positions = exclusive rank among the set's blocks (block counts that are no multiple of the kernels' 1024-block groups, empty and
    full sets), the packed buffer = index_select of the set's blocks, the check = the number of blocks OUTSIDE the set that hold
    anything but zeros (NaN and denormals count, -0.0 does not), unpack puts the blocks back and touches nothing else."""
    from uivr_amd.distributed import _positions
    native = uivr._native.native
    gen = torch.Generator().manual_seed(5 * block)
    stream = torch.cuda.current_stream().cuda_stream
    for n_blocks, density in ((1, 1.0), (5, 0.5), (1024, 0.3), (1025, 0.0), (3071, 1.0), (8195, 0.4)):
        mask = (torch.rand(n_blocks, generator=gen) < density).to(torch.uint8)
        body = torch.randn(n_blocks, block, generator=gen)
        outside = torch.nonzero(mask == 0).reshape(-1)
        body[outside] = 0                                              # a proper support ...
        extra = 0
        if outside.numel() >= 4:                                       # ... violated in three blocks
            body[outside[0], block - 1] = 1e-42
            body[outside[1], 3] = float("nan")
            body[outside[2], 0] = -0.0
            body[outside[3], block // 2] = -2.5
            extra = 3
        want_pos_cpu, want_cnt = _positions(mask)
        dmask, dbody = mask.to(gpu), body.to(gpu)
        pos, cnt = _positions(dmask)
        assert torch.equal(pos.cpu(), want_pos_cpu) and int(cnt) == int(want_cnt) == int(mask.sum())
        count = int(cnt)
        packed = torch.full((count * block + 1 + 4,), 7.0, device=gpu)             # [blocks | check | guard]
        packed[count * block] = 0.0
        native().grad_pack(stream, dbody.data_ptr(), pos.data_ptr(), n_blocks, block, packed.data_ptr(), packed.data_ptr() + 4 * count * block)
        src = torch.nonzero(mask).reshape(-1)
        assert torch.equal(packed[:count * block].cpu().view(-1, block), body.index_select(0, src))
        assert float(packed[count * block]) == float(extra)
        assert bool((packed[count * block + 1:] == 7.0).all())
        summed = packed.clone(); summed[:count * block] *= 2.0                     # "the all-reduce"
        flat = dbody.clone()
        native().grad_unpack(stream, summed.data_ptr(), pos.data_ptr(), n_blocks, block, flat.data_ptr())
        want = body.clone(); want[src] *= 2.0
        got = flat.cpu()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))           # bit for bit (NaN and -0.0 outside the set untouched)
    with pytest.raises(RuntimeError):
        native().grad_pack(stream, dbody.data_ptr(), pos.data_ptr(), 1, 96, packed.data_ptr(), packed.data_ptr())


def test_grad_pack_and_unpack_time_at_256(uivr, gpu):
    """The packing passes of a 256^3 gradient buffer (256 MiB, the headline's support: 44 % of the blocks) are two streaming kernels:
    pack reads the whole buffer once (the set's blocks are copied, the others checked), unpack writes the set's blocks back."""
    from uivr_amd import synthetic
    from uivr_amd.distributed import COMPACT_BLOCK_FLOATS as B, gradient_support
    native = uivr._native.native
    scene = synthetic.dust_devil_scene(res=256, film=64, device=gpu)
    grads = uivr.alloc_grads(scene)
    sup = gradient_support(scene.medium.sigma_t, grads)
    flat = grads["_flat"]
    n_blocks = flat.numel() // B
    count = sup.count
    assert 0.2 < count / n_blocks < 0.7
    packed = torch.zeros(count * B + 1, device=gpu)
    stream = torch.cuda.current_stream().cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    best = [1e9, 1e9]
    for _ in range(5):
        ev[0].record()
        native().grad_pack(stream, flat.data_ptr(), sup.pos.data_ptr(), n_blocks, B, packed.data_ptr(), packed.data_ptr() + 4 * count * B)
        ev[1].record()
        native().grad_unpack(stream, packed.data_ptr(), sup.pos.data_ptr(), n_blocks, B, flat.data_ptr())
        ev[2].record()
        torch.cuda.synchronize()
        best = [min(best[0], ev[0].elapsed_time(ev[1])), min(best[1], ev[1].elapsed_time(ev[2]))]
    print(f"grad_pack {best[0]:.3f} ms, grad_unpack {best[1]:.3f} ms at 256^3 ({count} of {n_blocks} blocks)")
    assert best[0] + best[1] < 0.4                                   # (measured 0.1 - 0.15 ms; the torch formulation: ~1 ms)


@pytest.mark.parametrize("spp", [1, 7, 32, 128, 200, 1024])
def test_film_develop_is_the_sample_mean(uivr, gpu, spp):
    """Box film (python/batched.py:176-197): image = mean over the pixel's samples - the thread-per-channel kernel
    (few samples) and the wave-per-pixel kernel (>= 128 samples: the optimisation loop's 1024) against float64."""
    scene = uivr.scene_to(uivr.cube_test_scene(8, 8), gpu)
    integ = uivr.load_dict({"type": "volpathsimple"})
    n_pix = 37
    gen = torch.Generator().manual_seed(spp)
    L = (torch.rand(n_pix * spp, 3, generator=gen) * 3.0).to(gpu)
    img = integ.develop(scene, L, spp)
    want = L.view(n_pix, spp, 3).double().mean(dim=1)
    assert tuple(img.shape) == (n_pix, 3)
    assert float((img.double() - want).abs().max()) <= 2e-6 * 3.0
    torch.testing.assert_close(img, integ.develop(scene, L, spp), rtol=0, atol=0)        # deterministic
