"""ctypes binding of the CPU ORACLE (oracle/drt_oracle.c) - test infrastructure.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It consumes the plain scene dataclasses of the product package by duck
typing (it never imports the product).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libdrt_oracle.so")
_lib = None


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "hide_emitters", "use_nee", "use_drt", "use_drt_subsampling", "use_drt_mis",
        "max_depth", "rr_depth")]


class Medium(C.Structure):
    _fields_ = [("sigma_t", C.POINTER(C.c_float)), ("albedo", C.POINTER(C.c_float)),
                ("res", C.c_int32 * 3), ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("scale", C.c_float), ("majorant_factor", C.c_int32), ("res_colour", C.c_int32 * 3)]


class Emitter(C.Structure):
    _fields_ = [("radiance", C.c_float * 3), ("pixels", C.POINTER(C.c_float)), ("width", C.c_int32),
                ("height", C.c_int32), ("to_world", C.c_float * 9), ("scale", C.c_float)]


def make_emitter(emitter):
    """drto_emitter of a scene emitter (ConstantEmitter / EnvmapEmitter); returns (struct, keepalive)."""
    if hasattr(emitter, "pixels"):
        pix = _f32(emitter.pixels)
        assert pix.ndim == 3 and pix.shape[2] == 3
        e = Emitter((C.c_float * 3)(0, 0, 0), _fp(pix), pix.shape[1], pix.shape[0],
                    (C.c_float * 9)(*emitter.to_world_flat()), float(emitter.scale))
        return e, pix
    e = Emitter((C.c_float * 3)(*emitter.radiance))
    return e, None


class Sensor(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("left", C.c_float * 3), ("up", C.c_float * 3),
                ("dir", C.c_float * 3), ("tan_x", C.c_float), ("tan_y", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_rays", "n_dt", "n_rt", "n_drt", "n_alb", "n_tr", "n_rt_adj", "n_sc", "n_sc_alb")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class NerfConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hide_emitters", "queries_per_ray", "jittering_enabled", "activation_relu")]


class Job(C.Structure):
    _fields_ = [("cfg", C.POINTER(Config)), ("medium", C.POINTER(Medium)),
                ("emitter", C.POINTER(Emitter)), ("sensor", C.POINTER(Sensor)),
                ("rays_o", C.POINTER(C.c_float)), ("rays_d", C.POINTER(C.c_float)),
                ("n_rays", C.c_uint64), ("ray_offset", C.c_uint64),
                ("spp", C.c_uint32), ("seed", C.c_uint32), ("n_threads", C.c_int32), ("grad_cache_log2", C.c_int32)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src_newer = (not os.path.exists(_LIB_PATH) or
                 any(os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
                     for f in ("drt_oracle.c", "drt_oracle.h", "Makefile")))
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.drto_render_primal.argtypes = [C.POINTER(Job), fp, C.POINTER(Counters)]
        L.drto_render_backward.argtypes = [C.POINTER(Job), fp, fp, dp, dp, C.POINTER(Counters)]
        L.drto_h1_step.argtypes = [C.POINTER(Job), fp, fp, dp, dp, dp, C.POINTER(Counters)]
        L.drto_render_textbook.argtypes = [C.POINTER(Job), fp]
        L.drto_nerf_render.argtypes = [C.POINTER(Job), C.POINTER(NerfConfig), fp, C.c_int, fp, fp, fp, dp, dp,
                                       C.POINTER(Counters)]
        L.drto_expf.argtypes = [C.c_float]
        L.drto_expf.restype = C.c_float
        L.drto_batch_sample_rays.argtypes = [C.POINTER(Sensor), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             fp, fp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.drto_batch_sample_rays.restype = None
        L.drto_tea32.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.drto_tea32.restype = C.c_uint32
        L.drto_pcg32_floats.argtypes = [C.c_uint32, C.c_uint32, C.c_int, fp]
        L.drto_pcg32_floats.restype = None
        L.drto_pcg32_raw.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
        L.drto_pcg32_raw.restype = None
        L.drto_uniform_sphere.argtypes = [C.c_float, C.c_float, fp]
        L.drto_uniform_sphere.restype = None
        L.drto_logf.argtypes = [C.c_float]
        L.drto_logf.restype = C.c_float
        L.drto_sincos_2pi.argtypes = [C.c_float, fp, fp]
        L.drto_sincos_2pi.restype = None
        L.drto_eval_sigma_t.argtypes = [C.POINTER(Medium), fp]
        L.drto_eval_sigma_t.restype = C.c_float
        L.drto_eval_albedo.argtypes = [C.POINTER(Medium), fp, fp]
        L.drto_eval_albedo.restype = None
        L.drto_majorant.argtypes = [C.POINTER(Medium)]
        L.drto_majorant.restype = C.c_float
        L.drto_majorant_grid.argtypes = [C.POINTER(Medium), C.POINTER(C.c_int32), fp]
        L.drto_majorant_grid.restype = C.c_int
        L.drto_ratio_tracking_mean.argtypes = [C.POINTER(Medium), fp, fp, C.c_float, C.c_uint32, C.c_int]
        L.drto_ratio_tracking_mean.restype = C.c_double
        L.drto_sample_interaction_drt.argtypes = [C.POINTER(Medium), fp, fp, C.c_uint32, C.c_uint32, C.c_int,
                                                  C.POINTER(C.c_int32), fp, fp]
        L.drto_sample_interaction_drt.restype = C.c_float
        L.drto_box_hit.argtypes = [C.POINTER(Medium), fp, fp, fp, fp]
        L.drto_box_hit.restype = C.c_int
        L.drto_sensor_ray.argtypes = [C.POINTER(Sensor), C.c_uint32, C.c_float, C.c_float, fp, fp]
        L.drto_sensor_ray.restype = None
        L.drto_alt_seed.argtypes = [C.c_uint32, C.c_int]
        L.drto_alt_seed.restype = C.c_uint32
        L.drto_atan2f.argtypes = [C.c_float, C.c_float]
        L.drto_atan2f.restype = C.c_float
        L.drto_envmap_eval.argtypes = [C.POINTER(Emitter), fp, fp]
        L.drto_envmap_pdf.argtypes = [C.POINTER(Emitter), fp]
        L.drto_envmap_pdf.restype = C.c_float
        L.drto_envmap_sample.argtypes = [C.POINTER(Emitter), C.c_float, C.c_float, fp, fp, fp]
        L.drto_envmap_tables.argtypes = [C.POINTER(Emitter), fp, fp]
        _lib = L
    return _lib


def _f32(a):
    if hasattr(a, "detach"):  # torch tensor
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_config(props: dict) -> Config:
    """props: the integrator property dict (volpathsimple.py:22-34 names)."""
    max_depth = int(props["max_depth"])
    return Config(
        hide_emitters=int(props.get("hide_emitters", False)),
        use_nee=int(props.get("use_nee", True)),
        use_drt=int(props.get("use_drt", True)),
        use_drt_subsampling=int(props.get("use_drt_subsampling", True)),
        use_drt_mis=int(props.get("use_drt_mis", True)),
        max_depth=max_depth,
        rr_depth=int(props.get("rr_depth", 5)))   # Mitsuba RBIntegrator default


class OracleScene:
    """Keeps the numpy buffers alive next to the C structs that point into them."""

    def __init__(self, scene, sensor_index: Optional[int] = 0):
        m = scene.medium
        self.sigma_t = _f32(m.sigma_t)
        # the colour grids (albedo; the nerf integrator's emission) may live on their own lattice (scene_config.py:108-110)
        em = getattr(m, "emission", None)
        cshape = tuple(m.albedo.shape[:3]) if m.albedo is not None else (tuple(em.shape[:3]) if em is not None else tuple(self.sigma_t.shape[:3]))
        self.albedo = _f32(m.albedo) if m.albedo is not None else np.zeros(cshape + (3,), np.float32)
        z, y, x = self.sigma_t.shape[:3]
        assert self.albedo.shape[-1] == 3
        own = cshape != (z, y, x)
        self.colour_shape = cshape
        self.medium = Medium(_fp(self.sigma_t), _fp(self.albedo), (C.c_int32 * 3)(x, y, z),
                             (C.c_float * 3)(*m.bbox_min), (C.c_float * 3)(*m.bbox_max),
                             float(m.scale), int(getattr(m, "majorant_resolution_factor", 0)),
                             (C.c_int32 * 3)(*(cshape[2], cshape[1], cshape[0]) if own else (0, 0, 0)))
        self.emitter, self._emitter_pixels = make_emitter(scene.emitter)
        self.sensor = None
        self.film = None
        if sensor_index is not None and scene.sensors:
            s = scene.sensors[sensor_index]
            f = s.frame()
            self.sensor = Sensor((C.c_float * 3)(*f["origin"]), (C.c_float * 3)(*f["left"]),
                                 (C.c_float * 3)(*f["up"]), (C.c_float * 3)(*f["dir"]),
                                 float(f["tan_x"]), float(f["tan_y"]), s.width, s.height)
            self.film = (s.width, s.height)

    def grid_shape(self):
        return self.sigma_t.shape[:3]

    def job(self, cfg: Config, spp: int, seed: int, n_rays=None, ray_offset=0,
            rays_o=None, rays_d=None, n_threads=0, grad_cache_log2=0) -> Job:
        self._cfg = cfg
        j = Job()
        j.cfg = C.pointer(cfg)
        j.medium = C.pointer(self.medium)
        j.emitter = C.pointer(self.emitter)
        if rays_o is None:
            assert self.sensor is not None
            j.sensor = C.pointer(self.sensor)
            total = self.film[0] * self.film[1] * spp
            j.n_rays = total - ray_offset if n_rays is None else n_rays
        else:
            self._ro, self._rd = _f32(rays_o), _f32(rays_d)
            j.rays_o, j.rays_d = _fp(self._ro), _fp(self._rd)
            j.n_rays = self._ro.shape[0] if n_rays is None else n_rays
        j.ray_offset = ray_offset
        j.spp = spp
        j.seed = seed
        j.n_threads = n_threads
        j.grad_cache_log2 = grad_cache_log2
        return j


def render_primal(oscene: OracleScene, props: dict, spp: int, seed: int, **kw):
    """-> (L [n,3] float32, counters dict)"""
    job = oscene.job(make_config(props), spp, seed, **kw)
    L = np.zeros((job.n_rays, 3), dtype=np.float32)
    cnt = Counters()
    rc = lib().drto_render_primal(C.byref(job), _fp(L), C.byref(cnt))
    if rc:
        raise RuntimeError(f"drto_render_primal failed: {rc}")
    return L, cnt.as_dict()


def render_backward(oscene: OracleScene, props: dict, spp: int, seed: int, dL, L_in, **kw):
    """-> (grad_sigma_t (Z,Y,X,1) f64, grad_albedo (Z,Y,X,3) f64, counters dict)"""
    job = oscene.job(make_config(props), spp, seed, **kw)
    dL, L_in = _f32(dL), _f32(L_in)
    assert dL.shape == (job.n_rays, 3) and L_in.shape == (job.n_rays, 3)
    z, y, x = oscene.grid_shape()
    gs = np.zeros((z, y, x, 1), dtype=np.float64)
    ga = np.zeros(tuple(oscene.colour_shape) + (3,), dtype=np.float64)
    cnt = Counters()
    rc = lib().drto_render_backward(C.byref(job), _fp(dL), _fp(L_in), _dp(gs), _dp(ga), C.byref(cnt))
    if rc:
        raise RuntimeError(f"drto_render_backward failed: {rc}")
    return gs, ga, cnt.as_dict()


def h1_step(oscene: OracleScene, props: dict, spp: int, seed: int, **kw):
    """One primal -> dL -> adjoint step with loss mean((img-0.5)^2).
    -> dict(image, loss, grad_sigma_t, grad_albedo, counters, L)"""
    job = oscene.job(make_config(props), spp, seed, **kw)
    n = job.n_rays
    L = np.zeros((n, 3), dtype=np.float32)
    image = np.zeros((n // spp, 3), dtype=np.float32)
    z, y, x = oscene.grid_shape()
    gs = np.zeros((z, y, x, 1), dtype=np.float64)
    ga = np.zeros(tuple(oscene.colour_shape) + (3,), dtype=np.float64)
    loss = C.c_double(0.0)
    cnt = Counters()
    rc = lib().drto_h1_step(C.byref(job), _fp(L), _fp(image), C.byref(loss), _dp(gs), _dp(ga),
                            C.byref(cnt))
    if rc:
        raise RuntimeError(f"drto_h1_step failed: {rc}")
    return dict(image=image, loss=loss.value, grad_sigma_t=gs, grad_albedo=ga,
                counters=cnt.as_dict(), L=L)


def make_nerf_config(props: dict) -> NerfConfig:
    """props: NeRFIntegrator properties (nerf.py:30-35)."""
    act = str(props.get("activation", "identity")).lower()
    assert act in ("identity", "relu")
    assert float(props.get("density_noise_std", 0.0)) == 0.0, "density noise is unsupported (nerf.py:160-162)"
    return NerfConfig(hide_emitters=int(props.get("hide_emitters", False)),
                      queries_per_ray=int(props.get("queries_per_ray", 128)),
                      jittering_enabled=int(props.get("jittering_enabled", True)),
                      activation_relu=int(act == "relu"))


def nerf_render(oscene: OracleScene, emission, props: dict, spp: int, seed: int, dL=None, L_in=None, **kw):
    """NeRFIntegrator.sample over the sensor's rays.  Primal (dL None): -> (L, counters).
    Backward: -> (grad_sigma_t, grad_emission, counters)."""
    cfg = Config(max_depth=0)
    job = oscene.job(cfg, spp, seed, **kw)
    ncfg = make_nerf_config(props)
    em = _f32(emission)
    cnt = Counters()
    # the colour lattice of THIS call is the emission grid's (the medium's albedo, which the march does not read, may have another one)
    z, y, x = oscene.grid_shape()
    ez, ey, ex = em.shape[:3]
    saved = tuple(oscene.medium.res_colour)
    oscene.medium.res_colour = (C.c_int32 * 3)(*((ex, ey, ez) if (ez, ey, ex) != (z, y, x) else (0, 0, 0)))
    try:
        return _nerf_render(oscene, job, ncfg, em, cnt, dL, L_in)
    finally:
        oscene.medium.res_colour = (C.c_int32 * 3)(*saved)


def _nerf_render(oscene, job, ncfg, em, cnt, dL, L_in):
    if dL is None:
        L = np.zeros((job.n_rays, 3), dtype=np.float32)
        rc = lib().drto_nerf_render(C.byref(job), C.byref(ncfg), _fp(em), 0, None, None, _fp(L), None, None, C.byref(cnt))
        if rc:
            raise RuntimeError(f"drto_nerf_render failed: {rc}")
        return L, cnt.as_dict()
    dL, L_in = _f32(dL), _f32(L_in)
    z, y, x = oscene.grid_shape()
    gs = np.zeros((z, y, x, 1), dtype=np.float64)
    ge = np.zeros(tuple(em.shape[:3]) + (3,), dtype=np.float64)
    rc = lib().drto_nerf_render(C.byref(job), C.byref(ncfg), _fp(em), 1, _fp(dL), _fp(L_in), None, _dp(gs), _dp(ge),
                                C.byref(cnt))
    if rc:
        raise RuntimeError(f"drto_nerf_render failed: {rc}")
    return gs, ge, cnt.as_dict()


def make_sensor(s) -> Sensor:
    f = s.frame()
    return Sensor((C.c_float * 3)(*f["origin"]), (C.c_float * 3)(*f["left"]), (C.c_float * 3)(*f["up"]),
                  (C.c_float * 3)(*f["dir"]), float(f["tan_x"]), float(f["tan_y"]), s.width, s.height)


def batch_sample_rays(sensors, batch_size: int, spp: int, sub_seed_pixels: int, sub_seed_rays: int):
    """sample_batch_pixels + sample_batch_rays (batched.py:397-467) -> (rays_o, rays_d, sensor_idx, pixels)."""
    arr = (Sensor * len(sensors))(*[make_sensor(s) for s in sensors])
    n = batch_size * spp
    ro = np.zeros((n, 3), np.float32); rd = np.zeros((n, 3), np.float32)
    si = np.zeros(batch_size, np.uint32); px = np.zeros((batch_size, 2), np.uint32)
    lib().drto_batch_sample_rays(arr, len(sensors), batch_size, spp, sub_seed_pixels, sub_seed_rays, _fp(ro), _fp(rd),
                                 si.ctypes.data_as(C.POINTER(C.c_uint32)), px.ctypes.data_as(C.POINTER(C.c_uint32)))
    return ro, rd, si, px


def render_textbook(oscene: OracleScene, props: dict, spp: int, seed: int, **kw):
    job = oscene.job(make_config(props), spp, seed, **kw)
    L = np.zeros((job.n_rays, 3), dtype=np.float32)
    rc = lib().drto_render_textbook(C.byref(job), _fp(L))
    if rc:
        raise RuntimeError(f"drto_render_textbook failed: {rc}")
    return L


def develop(L, spp: int):
    """box film: image[p] = mean over the pixel's spp samples (batched.py:176-197)."""
    L = np.asarray(L)
    return L.reshape(-1, spp, 3).mean(axis=1, dtype=np.float64).astype(np.float32)


def bytes_per_sample(cnt: dict, n_samples: int) -> float:
    """Algorithmic bytes per sample of one primal+adjoint step over `n_samples`
    camera rays: 60 B ray I/O + 32 B per sigma_t lookup + 96 B per
    albedo lookup + 64 B per sigma_t splat + 192 B per albedo splat."""
    n = max(1, n_samples)
    b = (60 * n + 32 * (cnt["n_dt"] + cnt["n_rt"] + cnt["n_drt"]) + 96 * cnt["n_alb"]
         + 64 * (cnt["n_tr"] + cnt["n_rt_adj"] + cnt["n_sc"]) + 192 * cnt["n_sc_alb"])
    return b / n


# ---- envmap emitter primitives (test hooks) ---------------------------------------------------
def envmap_eval(emitter, d):
    e, keep = make_emitter(emitter)
    dd = np.ascontiguousarray(d, dtype=np.float32)
    out = np.zeros(3, np.float32)
    lib().drto_envmap_eval(C.byref(e), _fp(dd), _fp(out))
    return out


def envmap_pdf(emitter, d):
    e, keep = make_emitter(emitter)
    dd = np.ascontiguousarray(d, dtype=np.float32)
    return float(lib().drto_envmap_pdf(C.byref(e), _fp(dd)))


def envmap_sample(emitter, u1, u2):
    """-> (direction[3], pdf, radiance / pdf [3])"""
    e, keep = make_emitter(emitter)
    d, w, pdf = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_float(0)
    rc = lib().drto_envmap_sample(C.byref(e), float(u1), float(u2), _fp(d), C.byref(pdf), _fp(w))
    assert rc == 0
    return d, float(pdf.value), w


def envmap_tables(emitter):
    """-> (marginal CDF [h+1], conditional CDFs [h, w+1])"""
    e, keep = make_emitter(emitter)
    marg = np.zeros(e.height + 1, np.float32)
    cond = np.zeros((e.height, e.width + 1), np.float32)
    rc = lib().drto_envmap_tables(C.byref(e), _fp(marg), _fp(cond))
    assert rc == 0
    return marg, cond


def sample_interaction_drt(oscene: OracleScene, o, d, seed: int, n: int, first: int = 0):
    """E2 test hook: n independent walks of Medium::sample_interaction_drt from `o` along `d` to the box
    exit -> (valid [n] bool, t' [n], W [n], maxt)."""
    o = _f32(np.asarray(o, dtype=np.float32)); d = _f32(np.asarray(d, dtype=np.float32))
    valid = np.zeros(n, dtype=np.int32); t = np.zeros(n, dtype=np.float32); W = np.zeros(n, dtype=np.float32)
    maxt = lib().drto_sample_interaction_drt(C.byref(oscene.medium), _fp(o), _fp(d), seed & 0xffffffff, first, n,
                                             valid.ctypes.data_as(C.POINTER(C.c_int32)), _fp(t), _fp(W))
    return valid.astype(bool), t, W, float(maxt)


def _fused_counters(cn: dict, cd: dict) -> dict:
    """events of both halves; the rays are the same rays, counted once"""
    out = {k: cn[k] + cd[k] for k in cd}
    out["n_rays"] = cd["n_rays"]
    return out


def fused_render_primal(oscene: OracleScene, drt_props: dict, nerf_props: dict, spp: int, seed: int, **kw):
    """BASELINE config 5, the fused nerf + volpathsimple pass, restated: NeRFIntegrator.sample (nerf.py:47-148) with
    emission = the medium's colour grid (albedo and emission are one asset, scene_config.py:109-110) and
    VolpathSimpleIntegrator.sample (volpathsimple.py:38-290) from the same rays and the same per-ray streams.
    -> (L [n, 6] = [nerf | volpathsimple], counters: the sum of both)."""
    Ln, cn = nerf_render(oscene, oscene.albedo, nerf_props, spp, seed, **kw)
    Ld, cd = render_primal(oscene, drt_props, spp, seed, **kw)
    return np.concatenate([Ln, Ld], axis=1), _fused_counters(cn, cd)


def fused_render_backward(oscene: OracleScene, drt_props: dict, nerf_props: dict, spp: int, seed: int, dL, L_in, **kw):
    """-> (grad_sigma_t, grad_rgb = albedo gradient + emission gradient, counters)."""
    dL, L_in = _f32(dL), _f32(L_in)
    gs_n, ge, cn = nerf_render(oscene, oscene.albedo, nerf_props, spp, seed, dL=np.ascontiguousarray(dL[:, :3]),
                               L_in=np.ascontiguousarray(L_in[:, :3]), **kw)
    gs_d, ga, cd = render_backward(oscene, drt_props, spp, seed, np.ascontiguousarray(dL[:, 3:]), np.ascontiguousarray(L_in[:, 3:]), **kw)
    return gs_n + gs_d, ge + ga, _fused_counters(cn, cd)
