import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle binding (oracle/binding.py); builds oracle/_build on demand."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def uivr():
    import uivr_amd
    return uivr_amd


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


DEFAULT_PROPS = dict(max_depth=64, rr_depth=1064, use_nee=True, use_drt=True, use_drt_subsampling=True,
                     use_drt_mis=True)

VARIANTS = {
    "drt": dict(use_drt=True, use_drt_subsampling=True, use_drt_mis=True),
    "drt-nomis": dict(use_drt=True, use_drt_subsampling=True, use_drt_mis=False),
    "quadratic": dict(use_drt=True, use_drt_subsampling=False, use_drt_mis=True),
    "quadratic-nomis": dict(use_drt=True, use_drt_subsampling=False, use_drt_mis=False),
    "basic": dict(use_drt=False),
}


def props_for(variant: str, **over):
    """Integrator properties of a registered estimator; Russian roulette disabled the way
    IntegratorConfig.create does it (rr_depth = max_depth + 1000, opt_config.py:105-106)
    unless `rr_depth` is given."""
    p = dict(max_depth=64, use_nee=True)
    p.update(VARIANTS[variant])
    p.update(over)
    p.setdefault("rr_depth", p["max_depth"] + 1000)
    return p
