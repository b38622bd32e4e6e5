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
