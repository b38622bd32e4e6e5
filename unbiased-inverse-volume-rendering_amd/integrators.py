"""Integrator plugins behind the reference's plugin surface.

Mirrors (names, argument meaning, error behaviour):
  * `mi.register_integrator("volpathsimple", lambda props: ...)`
    - python/integrators/volpathsimple.py:769
  * `mi.load_dict({'type': 'volpathsimple', ...})` - python/opt_config.py:108
  * `VolpathSimpleIntegrator.sample(mode, scene, sampler, ray, dL, state_in, active, **kwargs)
     -> (L, valid, state_out)` - python/integrators/volpathsimple.py:38-49
  * `integrator.aovs() -> []`, no `reparam` attribute - python/batched.py:152,223,235

The arithmetic runs in the HIP library (csrc/, C ABI in include/drt_hip.h) on the
device of the parameter tensors; nothing here computes radiance on the host.
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import IntEnum
from typing import Callable, Dict, Optional, Tuple

import torch

from ._native import native
from .scene import ALBEDO_KEY, EMISSION_KEY, SIGMA_T_KEY, PerspectiveSensor, Scene


class ADMode(IntEnum):
    """dr.ADMode values used by the reference (volpathsimple.py:51, batched.py:164,256,310)."""
    Primal = 0
    Forward = 1
    Backward = 2


# --- plugin registry (mi.register_integrator / mi.load_dict) ------------------------------
_INTEGRATORS: Dict[str, Callable[[dict], object]] = {}


def register_integrator(name: str, factory: Callable[[dict], object]) -> None:
    _INTEGRATORS[name] = factory


def load_dict(d: dict):
    """`mi.load_dict` for integrator dictionaries: {'type': <plugin>, **props}."""
    if "type" not in d:
        raise ValueError("load_dict: missing 'type'")
    t = d["type"]
    if t not in _INTEGRATORS:
        raise ValueError(f"load_dict: unknown integrator plugin '{t}' (registered: {sorted(_INTEGRATORS)})")
    props = {k: v for k, v in d.items() if k != "type"}
    return _INTEGRATORS[t](props)


# --- sampler / ray descriptors ---------------------------------------------------------------
class IndependentSampler:
    """`independent` sampler: one PCG32 per wavefront lane seeded with
    tea32(seed, lane) - only the seed lives on the host (batched.py:366-391)."""

    def __init__(self, seed: int = 0, sample_count: int = 1):
        self._seed = int(seed) & 0xffffffff
        self._sample_count = int(sample_count)

    def seed(self, seed: int, wavefront_size: Optional[int] = None) -> None:
        self._seed = int(seed) & 0xffffffff

    def clone(self) -> "IndependentSampler":
        return IndependentSampler(self._seed, self._sample_count)

    def set_sample_count(self, spp: int) -> None:
        self._sample_count = int(spp)

    def sample_count(self) -> int:
        return self._sample_count

    @property
    def seed_value(self) -> int:
        return self._seed


@dataclass
class RayBatch:
    """The `ray` argument of `sample()`.

    Either explicit rays (`o`, `d`: [n,3] float32 device tensors - the batched flow,
    batched.py:426-467) or rays generated on device from `sensor` (the `mi.render`
    flow: pixel = global_index // spp, film position drawn from the ray's stream).
    `ray_offset` / `interleave` place the local rays in the global wavefront so that
    a sharded render uses the same random streams as an unsharded one.
    """
    n_rays: int
    spp: int
    o: Optional[torch.Tensor] = None
    d: Optional[torch.Tensor] = None
    sensor: Optional[PerspectiveSensor] = None
    ray_offset: int = 0
    interleave: Optional[Tuple[int, int]] = None   # (chunk_rays, stride_rays)


def sample_tea_32(v0: int, v1: int, rounds: int = 4) -> Tuple[int, int]:
    """mi.sample_tea_32 (used for seeds: optimize.py:327-328, batched.py:119,411)."""
    v0 &= 0xffffffff
    v1 &= 0xffffffff
    s = 0
    for _ in range(rounds):
        s = (s + 0x9e3779b9) & 0xffffffff
        v0 = (v0 + ((((v1 << 4) & 0xffffffff) + 0xa341316c) ^ ((v1 + s) & 0xffffffff)
                    ^ ((v1 >> 5) + 0xc8013ea4))) & 0xffffffff
        v1 = (v1 + ((((v0 << 4) & 0xffffffff) + 0xad90777d) ^ ((v0 + s) & 0xffffffff)
                    ^ ((v0 >> 5) + 0x7e95761e))) & 0xffffffff
    return v0, v1


# --- the integrator ------------------------------------------------------------------------
class _DeviceIntegrator:
    """Shared plumbing of the integrator plugins: one native handle per device, medium / emitter /
    sensor binding, box film helpers."""

    param_keys = (SIGMA_T_KEY, ALBEDO_KEY)       # the differentiable grids this integrator reads
    needs_albedo = True

    def _native_props(self) -> dict:
        raise NotImplementedError

    # -- film helpers (hdrfilm + box filter, batched.py:176-197 / 298-306) -----
    def develop(self, scene: Scene, L: torch.Tensor, spp: int) -> torch.Tensor:
        h, dev = self._bind(scene)
        n_pix = L.shape[0] // spp
        img = torch.empty((n_pix, 3), dtype=torch.float32, device=dev)
        h.film_develop(L.data_ptr(), n_pix, int(spp), img.data_ptr())
        return img

    def film_backward(self, scene: Scene, grad_image: torch.Tensor, spp: int) -> torch.Tensor:
        h, dev = self._bind(scene)
        grad_image = grad_image.contiguous().view(-1, 3)
        n_pix = grad_image.shape[0]
        dL = torch.empty((n_pix * spp, 3), dtype=torch.float32, device=dev)
        h.film_backward(grad_image.data_ptr(), n_pix, int(spp), dL.data_ptr())
        return dL

    def native_handle(self, scene: Scene):
        return self._bind(scene)[0]

    def _colour_grid(self, scene: Scene):
        """The colour grid whose lattice the handle is told about (drt_set_colour_resolution): the albedo here, the emission for `nerf`."""
        return scene.medium.albedo

    def _bind(self, scene: Scene):
        m = scene.medium
        st, al = m.sigma_t, m.albedo
        if not isinstance(st, torch.Tensor) or (self.needs_albedo and not isinstance(al, torch.Tensor)):
            raise TypeError("the HIP integrator needs torch device tensors for the medium grids "
                            "(use scene_to(scene, device))")
        if not st.is_cuda:
            raise RuntimeError("sigma_t is not on a GPU: the integrator has no CPU path")
        dev = st.device
        _check(st, None, dev, "sigma_t")
        if st.dim() != 4 or st.shape[-1] != 1:
            raise ValueError(f"sigma_t must have shape (Z,Y,X,1), got {tuple(st.shape)}")
        if isinstance(al, torch.Tensor):
            _check(al, None, dev, "albedo")
            if al.dim() != 4 or al.shape[-1] != 3:
                raise ValueError(f"albedo must have shape (Z,Y,X,3), got {tuple(al.shape)}")
        # the colour grid this integrator reads (albedo; nerf: emission) may live on its OWN lattice, as every Mitsuba GridVolume does
        # (janga-smoke: 264 x 136 x 136 density, 256 x 128 x 128 albedo / emission, scene_config.py:108-110): drt_set_colour_resolution
        cg = self._colour_grid(scene)
        cshape = tuple(cg.shape[:3]) if isinstance(cg, torch.Tensor) else tuple(st.shape[:3])
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        if h is None:
            h = native(getattr(self, "test_hooks", False)).Integrator(self._native_props(), idx)
            self._handles[idx] = h
        h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        # The derived device state (apron-brick sigma_t copy, majorant, supergrid, empty-space mask) is
        # rebuilt by set_medium only.  The cache key is (address, version counter, geometry) of the bound
        # grids, and the entry HOLDS the bound tensors: while it is cached their storage cannot be freed,
        # so an equal address means the same storage (whose views share one version counter) - a freshly
        # allocated grid can never alias a cached key through the caching allocator.
        al_key = (al.data_ptr(), al._version) if isinstance(al, torch.Tensor) else (0, 0)
        key = (st.data_ptr(), st._version) + al_key + (tuple(st.shape), cshape,
               tuple(m.bbox_min), tuple(m.bbox_max), float(m.scale), int(m.majorant_resolution_factor))
        self._bind_emitter(h, idx, scene.emitter, dev)
        bound = self._bound.get(idx)
        if bound is None or bound[0] != key:
            z, y, x = st.shape[:3]
            h.set_medium(st.data_ptr(), al.data_ptr() if isinstance(al, torch.Tensor) else 0,
                         [int(x), int(y), int(z)],
                         [float(v) for v in m.bbox_min], [float(v) for v in m.bbox_max],
                         float(m.scale), int(m.majorant_resolution_factor))
            if cshape != tuple(st.shape[:3]):
                h.set_colour_resolution([int(cshape[2]), int(cshape[1]), int(cshape[0])])
            self._bound[idx] = (key, st, al)
        return h, dev

    def _bind_emitter(self, h, idx, emitter, dev):
        """`constant` or `envmap` (the only legal emitters, volpathsimple.py:16); the envmap upload
        builds the importance-sampling tables, so it is redone only when the map itself changes."""
        if hasattr(emitter, "pixels"):
            px = emitter.pixels
            if not isinstance(px, torch.Tensor):
                raise TypeError("envmap pixels must be a torch device tensor (use scene_to(scene, device))")
            if px.dim() != 3 or px.shape[-1] != 3:
                raise ValueError(f"envmap pixels must have shape (H, W, 3), got {tuple(px.shape)}")
            _check(px, tuple(px.shape), dev, "envmap pixels")
            R = emitter.to_world_flat()
            ekey = ("envmap", px.data_ptr(), px._version, tuple(px.shape), float(emitter.scale), tuple(R))
            bound = self._bound_emitter.get(idx)
            if bound is None or bound[0] != ekey:
                h.set_emitter_envmap(px.data_ptr(), int(px.shape[1]), int(px.shape[0]), R, float(emitter.scale))
                self._bound_emitter[idx] = (ekey, px)             # holds the map: see _bind
        else:
            ekey = ("constant",) + tuple(float(v) for v in emitter.radiance)
            bound = self._bound_emitter.get(idx)
            if bound is None or bound[0] != ekey:
                h.set_emitter_constant([float(v) for v in emitter.radiance])
                self._bound_emitter[idx] = (ekey, None)

    @staticmethod
    def _set_rays(h, ray: RayBatch):
        if ray.interleave:
            h.set_ray_interleave(int(ray.interleave[0]), int(ray.interleave[1]))
        else:
            h.set_ray_interleave(0, 0)
        if ray.o is None:
            if ray.sensor is None:
                raise ValueError("RayBatch needs explicit rays or a sensor")
            f = ray.sensor.frame()
            h.set_sensor_perspective([float(v) for v in f["origin"]], [float(v) for v in f["left"]],
                                     [float(v) for v in f["up"]], [float(v) for v in f["dir"]],
                                     float(f["tan_x"]), float(f["tan_y"]),
                                     int(ray.sensor.width), int(ray.sensor.height))

    @staticmethod
    def _ray_ptrs(ray: RayBatch, dev):
        n = int(ray.n_rays)
        if ray.o is not None:
            _check(ray.o, (n, 3), dev, "ray.o")
            _check(ray.d, (n, 3), dev, "ray.d")
            return n, ray.o.data_ptr(), ray.d.data_ptr()
        return n, 0, 0


class VolpathSimpleIntegrator(_DeviceIntegrator):
    """Differential-ratio-tracking volumetric path tracer (volpathsimple.py:10-36).

    Assumptions inherited from the reference: no surfaces, a single medium inside a
    convex (here: axis-aligned box) boundary with a null BSDF, one infinite emitter.
    """

    def __init__(self, props: Optional[dict] = None):
        props = dict(props or {})
        self.hide_emitters = bool(props.get("hide_emitters", False))
        self.use_nee = bool(props.get("use_nee", True))
        self.use_drt = bool(props.get("use_drt", True))
        self.use_drt_subsampling = bool(props.get("use_drt_subsampling", True))
        self.use_drt_mis = bool(props.get("use_drt_mis", True))
        # not a reference property: bind the library flavour with test hooks (kernel-variant selection, ablations)
        self.test_hooks = bool(props.get("test_hooks", False))
        # RBIntegrator base properties (defaults of mi.ad.integrators.common)
        self.max_depth = int(props.get("max_depth", 6))
        self.rr_depth = int(props.get("rr_depth", 5))
        if self.max_depth < 0:
            raise ValueError("max_depth must be >= 0 (unbounded depth is not supported)")
        self._handles: Dict[int, object] = {}
        self._bound: Dict[int, tuple] = {}
        self._bound_emitter: Dict[int, tuple] = {}

    # -- reference surface ---------------------------------------------------
    def aovs(self):
        return []

    def props(self) -> dict:
        return dict(hide_emitters=self.hide_emitters, use_nee=self.use_nee, use_drt=self.use_drt,
                    use_drt_subsampling=self.use_drt_subsampling, use_drt_mis=self.use_drt_mis,
                    max_depth=self.max_depth, rr_depth=self.rr_depth)

    def sample(self, mode, scene: Scene, sampler: IndependentSampler, ray: RayBatch,
               δL: Optional[torch.Tensor] = None, state_in: Optional[torch.Tensor] = None,
               active=None, grads: Optional[Dict[str, torch.Tensor]] = None, **kwargs):
        """-> (L, valid, state_out).  Primal: L = state_out = radiance [n,3].
        Backward: gradients are ACCUMULATED into `grads[key]` (tensors shaped like the
        parameters); returns (None, True, None).  Extra kwargs (`depth`, `reparam`)
        are absorbed like the reference does (volpathsimple.py:47)."""
        mode = ADMode(int(mode))
        h, dev = self._bind(scene)
        self._set_rays(h, ray)
        n, ro, rd = self._ray_ptrs(ray, dev)
        if mode == ADMode.Primal:
            L = torch.empty((n, 3), dtype=torch.float32, device=dev)
            h.render_primal(ro, rd, n, int(ray.ray_offset), int(ray.spp), sampler.seed_value, L.data_ptr())
            return L, True, L
        if mode == ADMode.Backward:
            if δL is None or state_in is None:
                raise ValueError("sample(Backward) needs δL and state_in")
            if grads is None:
                raise ValueError("sample(Backward) needs `grads` (dict of accumulation tensors)")
            _check(δL, (n, 3), dev, "δL")
            _check(state_in, (n, 3), dev, "state_in")
            gs, ga = grads[SIGMA_T_KEY], grads[ALBEDO_KEY]
            _check(gs, tuple(scene.medium.sigma_t.shape), dev, "grads[sigma_t]")
            _check(ga, tuple(scene.medium.albedo.shape), dev, "grads[albedo]")
            h.render_backward(ro, rd, n, int(ray.ray_offset), int(ray.spp), sampler.seed_value,
                              δL.data_ptr(), state_in.data_ptr(), gs.data_ptr(), ga.data_ptr())
            return None, True, None
        raise NotImplementedError("forward-mode differentiation is not supported "
                                  "(render_batch_forward raises in the reference too, batched.py:200-209)")

    def _native_props(self) -> dict:
        return self.props()


def _check(t: torch.Tensor, shape, dev, name: str):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if t.device != dev:
        raise ValueError(f"{name} is on {t.device}, expected {dev}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")


class NeRFIntegrator(_DeviceIntegrator):
    """Simplified NeRF-style integrator: emission accumulated along the ray, no scattering
    (python/integrators/nerf.py:20-35).  Reads `medium.sigma_t` and `medium.emission`."""

    param_keys = (SIGMA_T_KEY, EMISSION_KEY)
    needs_albedo = False

    def _colour_grid(self, scene: Scene):
        return scene.medium.emission

    def __init__(self, props: Optional[dict] = None):
        props = dict(props or {})
        self.hide_emitters = bool(props.get("hide_emitters", False))
        self.queries_per_ray = int(props.get("queries_per_ray", 128))
        self.density_noise_std = float(props.get("density_noise_std", 0.0))
        self.jittering_enabled = bool(props.get("jittering_enabled", True))
        self.activation_type = str(props.get("activation", "identity")).lower()
        self.test_hooks = bool(props.get("test_hooks", False))
        self.max_depth = int(props.get("max_depth", 6))          # RBIntegrator base; unused (nerf.py)
        self.rr_depth = int(props.get("rr_depth", 5))
        if self.activation_type not in ("identity", "relu"):
            raise ValueError(f"Unsupported activation: {self.activation_type}")         # nerf.py:44
        if self.density_noise_std > 0:
            raise NotImplementedError("density_noise_std > 0 is incorrect in the reference's adjoint "
                                      "(nerf.py:160-162) and is not supported")
        if self.queries_per_ray < 2:
            raise ValueError("queries_per_ray must be >= 2")
        self._handles: Dict[int, object] = {}
        self._bound: Dict[int, tuple] = {}
        self._bound_emitter: Dict[int, tuple] = {}

    def aovs(self):
        return []

    def props(self) -> dict:
        return dict(hide_emitters=self.hide_emitters, queries_per_ray=self.queries_per_ray,
                    jittering_enabled=self.jittering_enabled, activation=self.activation_type,
                    density_noise_std=self.density_noise_std)

    def _native_props(self) -> dict:
        return dict(max_depth=0)

    def _nerf_props(self) -> dict:
        return dict(hide_emitters=self.hide_emitters, queries_per_ray=self.queries_per_ray,
                    jittering_enabled=self.jittering_enabled, activation_relu=self.activation_type == "relu")

    def sample(self, mode, scene: Scene, sampler: IndependentSampler, ray: RayBatch,
               δL: Optional[torch.Tensor] = None, state_in: Optional[torch.Tensor] = None,
               active=None, grads: Optional[Dict[str, torch.Tensor]] = None, **kwargs):
        """-> (L, valid, state_out) (nerf.py:47-58); Backward accumulates into `grads`
        (keys sigma_t / emission)."""
        mode = ADMode(int(mode))
        h, dev = self._bind(scene)
        em = scene.medium.emission
        if not isinstance(em, torch.Tensor):
            raise TypeError("the nerf integrator needs medium.emission as a torch device tensor")
        _check(em, None, dev, "emission")
        if em.dim() != 4 or em.shape[-1] != 3:
            raise ValueError(f"emission must have shape (Z,Y,X,3), got {tuple(em.shape)}")
        self._set_rays(h, ray)
        n, ro, rd = self._ray_ptrs(ray, dev)
        if mode == ADMode.Primal:
            L = torch.empty((n, 3), dtype=torch.float32, device=dev)
            h.nerf_render_primal(self._nerf_props(), em.data_ptr(), ro, rd, n, int(ray.ray_offset), int(ray.spp),
                                 sampler.seed_value, L.data_ptr())
            return L, True, L
        if mode == ADMode.Backward:
            if δL is None or state_in is None or grads is None:
                raise ValueError("sample(Backward) needs δL, state_in and grads")
            _check(δL, (n, 3), dev, "δL")
            _check(state_in, (n, 3), dev, "state_in")
            gs, ge = grads[SIGMA_T_KEY], grads[EMISSION_KEY]
            _check(gs, tuple(scene.medium.sigma_t.shape), dev, "grads[sigma_t]")
            _check(ge, tuple(em.shape), dev, "grads[emission]")
            h.nerf_render_backward(self._nerf_props(), em.data_ptr(), ro, rd, n, int(ray.ray_offset), int(ray.spp),
                                   sampler.seed_value, δL.data_ptr(), state_in.data_ptr(), gs.data_ptr(), ge.data_ptr())
            return None, True, None
        raise NotImplementedError("forward-mode differentiation is not supported")


class FusedNerfDrtIntegrator(VolpathSimpleIntegrator):
    """BASELINE config 5: the `nerf` march (python/integrators/nerf.py) and `volpathsimple` scattering
    (python/integrators/volpathsimple.py) in ONE pass over one interleaved four-channel [sigma_t, r, g, b] grid.

    The reference's scenes bind one asset as the medium's albedo AND emission grid
    (python/scene_config.py:109-110), so the parameters are `sigma_t` and ONE colour grid (key `albedo`); the
    radiance comes back as [n, 6] = [nerf rgb | volpathsimple rgb], each half bit-identical to its stand-alone
    integrator, and the backward pass accumulates both integrators' gradients into the same two grids.
    Properties: those of `volpathsimple` plus the `nerf` ones (`queries_per_ray`, `jittering_enabled`,
    `activation`, `nerf_hide_emitters`)."""

    def __init__(self, props: Optional[dict] = None):
        props = dict(props or {})
        self.queries_per_ray = int(props.pop("queries_per_ray", 128))
        self.jittering_enabled = bool(props.pop("jittering_enabled", True))
        self.activation_type = str(props.pop("activation", "identity")).lower()
        self.nerf_hide_emitters = bool(props.pop("nerf_hide_emitters", False))
        if self.activation_type not in ("identity", "relu"):
            raise ValueError(f"Unsupported activation: {self.activation_type}")
        if self.queries_per_ray < 2:
            raise ValueError("queries_per_ray must be >= 2")
        super().__init__(props)

    def _nerf_props(self) -> dict:
        return dict(hide_emitters=self.nerf_hide_emitters, queries_per_ray=self.queries_per_ray,
                    jittering_enabled=self.jittering_enabled, activation_relu=self.activation_type == "relu")

    def nerf_props(self) -> dict:
        return dict(hide_emitters=self.nerf_hide_emitters, queries_per_ray=self.queries_per_ray,
                    jittering_enabled=self.jittering_enabled, activation=self.activation_type)

    def develop(self, scene: Scene, L: torch.Tensor, spp: int) -> torch.Tensor:
        if L.shape[-1] != 6:
            return super().develop(scene, L, spp)
        return torch.cat([super().develop(scene, L[:, :3].contiguous(), spp), super().develop(scene, L[:, 3:].contiguous(), spp)], dim=1)

    def film_backward(self, scene: Scene, grad_image: torch.Tensor, spp: int) -> torch.Tensor:
        if grad_image.shape[-1] != 6:
            return super().film_backward(scene, grad_image, spp)
        return torch.cat([super().film_backward(scene, grad_image[:, :3].contiguous(), spp),
                          super().film_backward(scene, grad_image[:, 3:].contiguous(), spp)], dim=1)

    def sample(self, mode, scene: Scene, sampler: IndependentSampler, ray: RayBatch,
               δL: Optional[torch.Tensor] = None, state_in: Optional[torch.Tensor] = None,
               active=None, grads: Optional[Dict[str, torch.Tensor]] = None, **kwargs):
        """-> (L [n, 6], valid, state_out); Backward: δL / state_in are [n, 6], gradients accumulate into
        grads[sigma_t] and grads[albedo] (= the colour grid: albedo and emission are one parameter)."""
        mode = ADMode(int(mode))
        h, dev = self._bind(scene)
        self._set_rays(h, ray)
        n, ro, rd = self._ray_ptrs(ray, dev)
        if mode == ADMode.Primal:
            Ln = torch.empty((n, 3), dtype=torch.float32, device=dev)
            Ld = torch.empty((n, 3), dtype=torch.float32, device=dev)
            h.fused_render_primal(self._nerf_props(), ro, rd, n, int(ray.ray_offset), int(ray.spp), sampler.seed_value,
                                  Ln.data_ptr(), Ld.data_ptr())
            L = torch.cat([Ln, Ld], dim=1)
            return L, True, L
        if mode == ADMode.Backward:
            if δL is None or state_in is None or grads is None:
                raise ValueError("sample(Backward) needs δL, state_in and grads")
            _check(δL, (n, 6), dev, "δL")
            _check(state_in, (n, 6), dev, "state_in")
            gs, ga = grads[SIGMA_T_KEY], grads[ALBEDO_KEY]
            _check(gs, tuple(scene.medium.sigma_t.shape), dev, "grads[sigma_t]")
            _check(ga, tuple(scene.medium.albedo.shape), dev, "grads[albedo]")
            dLn, dLd = δL[:, :3].contiguous(), δL[:, 3:].contiguous()
            Ln, Ld = state_in[:, :3].contiguous(), state_in[:, 3:].contiguous()
            h.fused_render_backward(self._nerf_props(), ro, rd, n, int(ray.ray_offset), int(ray.spp), sampler.seed_value,
                                    dLn.data_ptr(), Ln.data_ptr(), dLd.data_ptr(), Ld.data_ptr(), gs.data_ptr(), ga.data_ptr())
            return None, True, None
        raise NotImplementedError("forward-mode differentiation is not supported")


register_integrator("nerf+volpathsimple", lambda props: FusedNerfDrtIntegrator(props))
register_integrator("volpathsimple", lambda props: VolpathSimpleIntegrator(props))
register_integrator("nerf", lambda props: NeRFIntegrator(props))
