"""bayesfast_amd: MI355X-native implementation of BayesFast's data-parallel hot path.

Polynomial-surrogate logp/grad, leapfrog, NUTS/HMC over many chains and the surrogate's least-squares fit
run as hand-written gfx950 HIP kernels behind a C ABI (include/bfhip.h); this package is the thin Python
host side that mirrors the reference's ``modules``/``samplers`` interfaces for that path.
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
