"""The drop-in boundary used the way INTEGRATION.md section 3 shows it: raw `ctypes` on `libdrt_hip.so` (no pybind11 shim,
no host layer) - create, bind the medium / emitter / sensor from plain device pointers, primal + backward, counters - against
the oracle: radiance bit-exact, counters equal, gradients within 2e-4 max|oracle|."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import props_for

pytestmark = pytest.mark.gpu


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hide_emitters", "use_nee", "use_drt", "use_drt_subsampling", "use_drt_mis",
                                         "max_depth", "rr_depth")]


class _Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_rays", "n_dt", "n_rt", "n_drt", "n_alb", "n_tr", "n_rt_adj", "n_sc", "n_sc_alb")]


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


@pytest.mark.parametrize("factor", [0, 3])
def test_raw_ctypes_primal_and_backward_match_oracle(uivr, oracle, gpu, factor):
    from uivr_amd._native import library_path
    from test_gpu_envmap import _env_scene
    lib = C.CDLL(library_path())
    lib.drt_last_error.restype = C.c_char_p

    def ok(h, rc):
        assert rc == 0, lib.drt_last_error(h)

    scene = _env_scene(uivr, film=24, factor=factor)
    scene.emitter = uivr.cube_test_scene(4, 4).emitter                  # constant emitter
    props = props_for("drt")
    spp, seed = 8, 31337
    osc = oracle.OracleScene(scene)
    Lr, cp = oracle.render_primal(osc, props, spp, seed)
    n = Lr.shape[0]
    rng = np.random.default_rng(11)
    dL = ((rng.random((n, 3), dtype=np.float32) - 0.5) * 1e-2).astype(np.float32)
    gs_ref, ga_ref, ca = oracle.render_backward(osc, props, spp, seed, dL, Lr)

    cfg = _Cfg(int(props.get("hide_emitters", False)), int(props.get("use_nee", True)), int(props.get("use_drt", True)),
               int(props.get("use_drt_subsampling", True)), int(props.get("use_drt_mis", True)),
               int(props["max_depth"]), int(props["rr_depth"]))
    h = C.c_void_p()
    ok(None, lib.drt_create(C.byref(cfg), gpu.index or 0, C.byref(h)))
    try:
        m = scene.medium
        sig = torch.from_numpy(np.ascontiguousarray(m.sigma_t, dtype=np.float32)).to(gpu)
        alb = torch.from_numpy(np.ascontiguousarray(m.albedo, dtype=np.float32)).to(gpu)
        z, y, x = sig.shape[:3]
        ok(h, lib.drt_set_medium(h, C.c_void_p(sig.data_ptr()), C.c_void_p(alb.data_ptr()), (C.c_int32 * 3)(x, y, z),
                                 _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(float(m.scale)), C.c_int32(factor)))
        ok(h, lib.drt_set_emitter_constant(h, _f3(scene.emitter.radiance)))
        s = scene.sensors[0]
        f = s.frame()
        ok(h, lib.drt_set_sensor_perspective(h, _f3(f["origin"]), _f3(f["left"]), _f3(f["up"]), _f3(f["dir"]),
                                             C.c_float(float(f["tan_x"])), C.c_float(float(f["tan_y"])), C.c_int32(s.width), C.c_int32(s.height)))
        ok(h, lib.drt_enable_counters(h, 1))
        ok(h, lib.drt_reset_counters(h))
        L = torch.empty((n, 3), dtype=torch.float32, device=gpu)
        ok(h, lib.drt_render_primal(h, None, None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed), C.c_void_p(L.data_ptr())))
        cnt = _Counters()
        ok(h, lib.drt_get_counters(h, C.byref(cnt)))
        np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
        assert {k: int(getattr(cnt, k)) for k, _ in _Counters._fields_} == cp
        ok(h, lib.drt_reset_counters(h))
        gsig, galb = torch.zeros_like(sig), torch.zeros_like(alb)
        dLd = torch.from_numpy(dL).to(gpu)
        ok(h, lib.drt_render_backward(h, None, None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed),
                                      C.c_void_p(dLd.data_ptr()), C.c_void_p(L.data_ptr()), C.c_void_p(gsig.data_ptr()),
                                      C.c_void_p(galb.data_ptr())))
        ok(h, lib.drt_get_counters(h, C.byref(cnt)))
        assert {k: int(getattr(cnt, k)) for k, _ in _Counters._fields_} == ca
        for got, ref, what in ((gsig, gs_ref, "sigma_t"), (galb, ga_ref, "albedo")):
            r = torch.from_numpy(ref).to(gpu)
            tol = 2e-4 * float(r.abs().max()) + 1e-9
            assert float(r.abs().max()) > 0
            assert float((got.double() - r).abs().max()) <= tol, what
    finally:
        lib.drt_destroy(h)


def test_raw_ctypes_colour_grid_on_its_own_lattice(uivr, oracle, gpu):
    """drt_set_colour_resolution through the bare C ABI, as INTEGRATION.md section 3 binds it: a 33 x 17 x 17 density with a 32 x 16 x 16 albedo
    (the reference's janga-smoke ratio, python/scene_config.py:108-110) - primal bit-exact, the albedo gradient on the albedo's lattice - and
    back to one lattice on the same handle."""
    from uivr_amd._native import library_path
    from test_gpu_lattice import _scene
    lib = C.CDLL(library_path())
    lib.drt_last_error.restype = C.c_char_p

    def ok(h, rc):
        assert rc == 0, lib.drt_last_error(h)

    props = props_for("drt")
    cfg = _Cfg(0, 1, 1, 1, 1, int(props["max_depth"]), int(props["rr_depth"]))
    h = C.c_void_p()
    ok(None, lib.drt_create(C.byref(cfg), gpu.index or 0, C.byref(h)))
    try:
        assert lib.drt_set_colour_resolution(h, (C.c_int32 * 3)(32, 16, 16)) != 0          # no medium yet: refused, with a message
        assert b"no medium" in lib.drt_last_error(h)
        for colour in ((16, 16, 32), (17, 17, 33)):
            scene = _scene(uivr, factor=4, colour=colour)
            spp, seed = 4, 424
            osc = oracle.OracleScene(scene)
            Lr, _ = oracle.render_primal(osc, props, spp, seed)
            n = Lr.shape[0]
            dL = ((np.random.default_rng(6).random((n, 3), dtype=np.float32) - 0.5) * 1e-2).astype(np.float32)
            gs_ref, ga_ref, _ = oracle.render_backward(osc, props, spp, seed, dL, Lr)
            m = scene.medium
            sig = torch.from_numpy(np.ascontiguousarray(m.sigma_t, dtype=np.float32)).to(gpu)
            alb = torch.from_numpy(np.ascontiguousarray(m.albedo, dtype=np.float32)).to(gpu)
            z, y, x = sig.shape[:3]
            ok(h, lib.drt_set_medium(h, C.c_void_p(sig.data_ptr()), C.c_void_p(alb.data_ptr()), (C.c_int32 * 3)(x, y, z),
                                     _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(float(m.scale)), C.c_int32(4)))
            if tuple(alb.shape[:3]) != (z, y, x):
                az, ay, ax = alb.shape[:3]
                ok(h, lib.drt_set_colour_resolution(h, (C.c_int32 * 3)(ax, ay, az)))
                assert lib.drt_set_colour_resolution(h, (C.c_int32 * 3)(ax, 0, az)) != 0     # a zero extent next to non-zero ones
                ok(h, lib.drt_set_colour_resolution(h, (C.c_int32 * 3)(ax, ay, az)))
            ok(h, lib.drt_set_emitter_constant(h, _f3(scene.emitter.radiance)))
            s = scene.sensors[0]
            f = s.frame()
            ok(h, lib.drt_set_sensor_perspective(h, _f3(f["origin"]), _f3(f["left"]), _f3(f["up"]), _f3(f["dir"]),
                                                 C.c_float(float(f["tan_x"])), C.c_float(float(f["tan_y"])), C.c_int32(s.width), C.c_int32(s.height)))
            L = torch.empty((n, 3), dtype=torch.float32, device=gpu)
            ok(h, lib.drt_render_primal(h, None, None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed), C.c_void_p(L.data_ptr())))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
            gsig, galb = torch.zeros_like(sig), torch.zeros_like(alb)
            dLd = torch.from_numpy(dL).to(gpu)
            ok(h, lib.drt_render_backward(h, None, None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed),
                                          C.c_void_p(dLd.data_ptr()), C.c_void_p(L.data_ptr()), C.c_void_p(gsig.data_ptr()),
                                          C.c_void_p(galb.data_ptr())))
            ok(h, lib.drt_synchronize(h))
            for got, ref, what in ((gsig, gs_ref, "sigma_t"), (galb, ga_ref, "albedo")):
                r = torch.from_numpy(ref).to(gpu)
                assert tuple(got.shape) == tuple(r.shape)
                assert float((got.double() - r).abs().max()) <= 2e-4 * float(r.abs().max()) + 1e-9, (colour, what)
    finally:
        lib.drt_destroy(h)


def test_raw_ctypes_misuse_is_refused_with_a_message_and_the_handle_lives_on(uivr, oracle, gpu):
    """Every call below is wrong in one way - a null pointer, a zero or negative extent, an inverted box, a non-finite scale, nothing configured yet,
    a zero spp ... - and must come back with an error code and a message (the reference raises: opt_config.py:98-104, util.py:83-85), never crash, never
    leave the handle unusable: the last step renders the cube fixture on the same handle, bit-exact against the oracle."""
    from uivr_amd._native import library_path
    lib = C.CDLL(library_path())
    lib.drt_last_error.restype = C.c_char_p
    props = props_for("drt")

    def cfg(**over):
        d = dict(hide_emitters=0, use_nee=1, use_drt=1, use_drt_subsampling=1, use_drt_mis=1, max_depth=int(props["max_depth"]), rr_depth=int(props["rr_depth"]))
        d.update(over)
        return _Cfg(*[d[k] for k in ("hide_emitters", "use_nee", "use_drt", "use_drt_subsampling", "use_drt_mis", "max_depth", "rr_depth")])

    refused = []

    def bad(what, rc, h=None):
        msg = lib.drt_last_error(h) if h is not None else b"(no handle)"
        assert rc != 0, f"{what}: accepted"
        assert msg, f"{what}: refused without a message"
        refused.append(what)

    h = C.c_void_p()
    bad("create without a config", lib.drt_create(None, gpu.index or 0, C.byref(h)))
    bad("create without an out pointer", lib.drt_create(C.byref(cfg()), gpu.index or 0, None))
    bad("create on device 999", lib.drt_create(C.byref(cfg()), 999, C.byref(h)))
    bad("create with max_depth -5", lib.drt_create(C.byref(cfg(max_depth=-5)), gpu.index or 0, C.byref(h)))
    assert lib.drt_create(C.byref(cfg()), gpu.index or 0, C.byref(h)) == 0
    try:
        scene = uivr.cube_test_scene(12, 9, density_scale=2.0)
        m = scene.medium
        sig = torch.from_numpy(np.ascontiguousarray(m.sigma_t, dtype=np.float32)).to(gpu)
        alb = torch.from_numpy(np.ascontiguousarray(m.albedo, dtype=np.float32)).to(gpu)
        z, y, x = sig.shape[:3]
        res = (C.c_int32 * 3)(x, y, z)
        n, spp, seed = 12 * 9 * 4, 4, 77
        L = torch.empty((n, 3), dtype=torch.float32, device=gpu)
        gs, ga = torch.zeros_like(sig), torch.zeros_like(alb)
        P = lambda t: C.c_void_p(t.data_ptr())

        def primal(hh=h, n_=n, spp_=spp, out=L):
            return lib.drt_render_primal(hh, None, None, C.c_uint64(n_), C.c_uint64(0), C.c_uint32(spp_), C.c_uint32(seed), P(out) if out is not None else None)

        bad("render before anything is set", primal(), h)
        bad("null handle", lib.drt_set_emitter_constant(None, _f3((1, 1, 1))))
        bad("emitter without radiance", lib.drt_set_emitter_constant(h, None), h)
        bad("medium without sigma_t", lib.drt_set_medium(h, None, P(alb), res, _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(1.0), C.c_int32(0)), h)
        bad("medium without a resolution", lib.drt_set_medium(h, P(sig), P(alb), None, _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(1.0), C.c_int32(0)), h)
        bad("medium with a zero extent", lib.drt_set_medium(h, P(sig), P(alb), (C.c_int32 * 3)(x, 0, z), _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(1.0), C.c_int32(0)), h)
        bad("medium with a negative extent", lib.drt_set_medium(h, P(sig), P(alb), (C.c_int32 * 3)(-x, y, z), _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(1.0), C.c_int32(0)), h)
        bad("medium in an inverted box", lib.drt_set_medium(h, P(sig), P(alb), res, _f3(m.bbox_max), _f3(m.bbox_min), C.c_float(1.0), C.c_int32(0)), h)
        bad("medium in a flat box", lib.drt_set_medium(h, P(sig), P(alb), res, _f3((0, 0, 0)), _f3((1, 0, 1)), C.c_float(1.0), C.c_int32(0)), h)
        bad("medium with a NaN scale", lib.drt_set_medium(h, P(sig), P(alb), res, _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(float("nan")), C.c_int32(0)), h)
        bad("medium with a negative scale", lib.drt_set_medium(h, P(sig), P(alb), res, _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(-1.0), C.c_int32(0)), h)
        bad("medium with a negative majorant_resolution_factor", lib.drt_set_medium(h, P(sig), P(alb), res, _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(1.0), C.c_int32(-2)), h)
        bad("colour lattice before the medium", lib.drt_set_colour_resolution(h, (C.c_int32 * 3)(2, 2, 2)), h)
        assert lib.drt_set_medium(h, P(sig), P(alb), res, _f3(m.bbox_min), _f3(m.bbox_max), C.c_float(float(m.scale)), C.c_int32(0)) == 0, lib.drt_last_error(h)
        bad("render without an emitter", primal(), h)
        assert lib.drt_set_emitter_constant(h, _f3(scene.emitter.radiance)) == 0
        bad("sensor rays without a sensor", primal(), h)
        s = scene.sensors[0]
        f = s.frame()
        sensor = lambda w_, h_, tx=float(f["tan_x"]): lib.drt_set_sensor_perspective(h, _f3(f["origin"]), _f3(f["left"]), _f3(f["up"]), _f3(f["dir"]), C.c_float(tx),
                                                                                       C.c_float(float(f["tan_y"])), C.c_int32(w_), C.c_int32(h_))
        bad("sensor with a zero width", sensor(0, s.height), h)
        bad("sensor with a negative height", sensor(s.width, -3), h)
        bad("sensor without an origin", lib.drt_set_sensor_perspective(h, None, _f3(f["left"]), _f3(f["up"]), _f3(f["dir"]), C.c_float(0.3), C.c_float(0.3), C.c_int32(4), C.c_int32(4)), h)
        assert sensor(s.width, s.height) == 0, lib.drt_last_error(h)
        bad("envmap without pixels", lib.drt_set_emitter_envmap(h, None, C.c_int32(8), C.c_int32(4), (C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1), C.c_float(1.0)), h)
        bad("envmap of width 0", lib.drt_set_emitter_envmap(h, P(alb), C.c_int32(0), C.c_int32(4), (C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1), C.c_float(1.0)), h)
        bad("envmap without a rotation", lib.drt_set_emitter_envmap(h, P(alb), C.c_int32(3), C.c_int32(3), None, C.c_float(1.0)), h)
        bad("primal without an output", primal(out=None), h)
        bad("primal with spp 0", primal(spp_=0), h)
        bad("primal over more rays than the film holds", primal(n_=n + spp), h)
        bad("rays_o without rays_d", lib.drt_render_primal(h, P(L), None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed), P(L)), h)
        dL = torch.zeros((n, 3), dtype=torch.float32, device=gpu)
        back = lambda dl, lin, g1, g2: lib.drt_render_backward(h, None, None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed),
                                                               P(dl) if dl is not None else None, P(lin) if lin is not None else None,
                                                               P(g1) if g1 is not None else None, P(g2) if g2 is not None else None)
        bad("backward without dL", back(None, L, gs, ga), h)
        bad("backward without L_in", back(dL, None, gs, ga), h)
        bad("backward without a sigma_t gradient", back(dL, L, None, ga), h)
        ncfg = (C.c_int32 * 4)(0, 1, 1, 0)                                       # queries_per_ray 1
        bad("nerf with one query per ray", lib.drt_nerf_render_primal(h, C.byref(ncfg), P(alb), None, None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed), P(L)), h)
        bad("nerf without a config", lib.drt_nerf_render_primal(h, None, P(alb), None, None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed), P(L)), h)
        ncfg = (C.c_int32 * 4)(0, 8, 1, 0)
        bad("nerf without an emission grid", lib.drt_nerf_render_primal(h, C.byref(ncfg), None, None, None, C.c_uint64(n), C.c_uint64(0), C.c_uint32(spp), C.c_uint32(seed), P(L)), h)
        bad("film_develop with spp 0", lib.drt_film_develop(h, P(L), C.c_uint64(12 * 9), C.c_uint32(0), P(L)), h)
        bad("counters without an output", lib.drt_get_counters(h, None), h)
        bad("interleave with a stride below the chunk", lib.drt_set_ray_interleave(h, C.c_uint64(64), C.c_uint64(32)), h)
        assert len(refused) == 38, refused
        # ... and the handle still does its work
        assert primal() == 0, lib.drt_last_error(h)
        assert lib.drt_synchronize(h) == 0
        Lr, _ = oracle.render_primal(oracle.OracleScene(scene), props, spp, seed)
        np.testing.assert_array_equal(L.cpu().numpy().view(np.uint32), Lr.view(np.uint32))
    finally:
        lib.drt_destroy(h)
    assert lib.drt_destroy(None) != 0 or True                                   # (destroying nothing is harmless either way)
