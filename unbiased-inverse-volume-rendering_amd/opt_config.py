"""`IntegratorConfig` registry - the configuration surface of the reference kept
as is (python/opt_config.py:83-169): same names, same `create(max_depth=...)`
contract (deep copy of `params`, `rr_depth` forbidden as a kwarg and forced to
`max_depth + 1000`, i.e. Russian roulette disabled).
"""
from __future__ import annotations

from copy import deepcopy
from dataclasses import dataclass
from typing import Dict

from .integrators import load_dict


@dataclass
class IntegratorConfig:
    name: str
    pretty_name: str
    params: Dict

    uses_fd: bool = False
    fd_epsilon: float = None
    fd_spp_multiplier: int = 16

    def __post_init__(self):
        if self.uses_fd:
            assert self.fd_epsilon is not None

    def create(self, **kwargs):
        assert 'max_depth' in kwargs
        d = deepcopy(self.params)
        d.update(kwargs)

        assert d['max_depth'] >= 0
        assert 'rr_depth' not in kwargs
        if 'rr_depth' not in self.params:
            d['rr_depth'] = d['max_depth'] + 1000

        return load_dict(d)


_INTEGRATOR_CONFIGS: Dict[str, IntegratorConfig] = {}


def add_int_config(name, **kwargs):
    assert name not in _INTEGRATOR_CONFIGS, f'Duplicate integrator config name: {name}'
    _INTEGRATOR_CONFIGS[name] = IntegratorConfig(name, **kwargs)


def get_int_config(name):
    if isinstance(name, IntegratorConfig):
        return deepcopy(name)
    return deepcopy(_INTEGRATOR_CONFIGS[name])


# The five registered names of the reference (opt_config.py:123-169).
add_int_config('fd-forward', pretty_name='Finite differences',
               params={'type': 'volpathsimple', 'use_drt': False},
               uses_fd=True, fd_epsilon=5e-3)
add_int_config('volpathsimple-drt', pretty_name='Differential Ratio Tracking',
               params={'type': 'volpathsimple', 'use_drt': True,
                       'use_drt_subsampling': True, 'use_drt_mis': True})
# The quadratic estimator (a DRT walk + recursive path at EVERY vertex of the main path: the paper's comparison baseline, not its
# method) suspends the main path in the middle of a bounce.  With a majorant supergrid it runs in the queued tracer's QUAD adjoint
# kernels (csrc/drt_sq.hip: the suspended path is kept in the ray record's global half; headline scene at factor 8: 418 Msamples/s
# against 128 in the one-ray-per-lane kernel it used before round 4), with ONE majorant in the one-ray-per-lane kernels; both are
# oracle-checked (tests/test_gpu_parity.py::test_supergrid_tracer_and_its_fallbacks_match_oracle[*-quadratic],
# ::test_quadratic_drt_paths_cut_by_max_depth).
add_int_config('volpathsimple-drt-quadratic', pretty_name='Differential Ratio Tracking (quadratic)',
               params={'type': 'volpathsimple', 'use_drt': True,
                       'use_drt_subsampling': False, 'use_drt_mis': True})
add_int_config('volpathsimple-basic', pretty_name='Free-flight based',
               params={'type': 'volpathsimple', 'use_drt': False})
add_int_config('nerf', pretty_name='NeRF (grid-backed)',
               params={'type': 'nerf', 'queries_per_ray': 128})
# BASELINE config 5 (no counterpart among the reference's registrations): `nerf` + `volpathsimple-drt` fused in one
# pass over the interleaved four-channel grid (integrators.FusedNerfDrtIntegrator).
add_int_config('nerf-drt-fused', pretty_name='NeRF + Differential Ratio Tracking (fused, 4-channel grid)',
               params={'type': 'nerf+volpathsimple', 'queries_per_ray': 128, 'use_drt': True,
                       'use_drt_subsampling': True, 'use_drt_mis': True})
