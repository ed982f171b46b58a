"""bayesfast_amd: MI355X-native implementation of BayesFast's data-parallel hot path.

Polynomial-surrogate logp/grad, leapfrog, NUTS/HMC over many chains and the surrogate's least-squares fit
run as hand-written gfx950 HIP kernels behind a C ABI (include/bfhip.h); this package is the thin Python
host side that mirrors the reference's ``modules`` / ``samplers`` / ``core.sample`` interfaces for that path.
"""
from . import _lib  # noqa: F401
from .modules import PolyConfig, PolyModel
from .core.module import Surrogate
from .core.density import SurrogateDensity, Chi2PipelineDensity, GaussianLink
from .core.sample import sample
from .samplers import NTrace, HTrace, TNTrace, GaussianBase, TraceTuple
from .utils import SystematicResampler
from .core.refit import select_fit_points, importance_weights
from .transforms import SIT
from .evidence import GBS, bridge

__all__ = ['PolyConfig', 'PolyModel', 'Surrogate', 'SurrogateDensity', 'Chi2PipelineDensity', 'GaussianLink', 'sample', 'NTrace', 'HTrace', 'TNTrace', 'GaussianBase', 'TraceTuple',
           'SystematicResampler', 'select_fit_points', 'importance_weights', 'SIT', 'GBS', 'bridge']
