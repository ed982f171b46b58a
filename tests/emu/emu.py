"""ctypes front end of the host emulation of the group sampler kernel (TEST INFRASTRUCTURE ONLY).

``tests/emu/_build/libbf_emu.so`` is ``bayesfast_amd/csrc/bfhip_group.h`` compiled for the host with one fibre per
lane (``emu_group.cpp``).  It lets the CPU test suite run the kernel's per-lane state machines, its cross-wave
reductions and its barrier placement against the oracle.  Nothing under ``bayesfast_amd/`` imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get('BF_EMU_LIB') or os.path.join(_HERE, '_build', 'libbf_emu.so')  # BF_EMU_LIB: the sanitizer build
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.environ.get('BF_EMU_LIB'):
            subprocess.check_call(['make', '-C', _HERE, '-s'])
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class EmuChains:
    """Same role as bayesfast_amd.chains.DeviceChains, on host arrays."""

    def __init__(self, spec, x_0, seed=0, first_stream=0, step_size=1., initial_weight=10., adapt_window=60):
        from bayesfast_amd.device import density_desc_from_spec
        from bayesfast_amd import _lib as bl
        self.bl = bl
        self.ds, self._keep = density_desc_from_spec(spec)
        x_0 = np.ascontiguousarray(x_0, dtype=np.float64)
        self.n_chain, self.d = x_0.shape
        self.rng = np.zeros((self.n_chain, 4), dtype=np.uint64)
        self.sc = np.zeros((self.n_chain, bl.SC_N))
        self.vec = np.zeros((self.n_chain, bl.VEC_N, self.d))
        self.n_leapfrog = np.zeros(1, dtype=np.uint64)
        L = lib()
        L.bfemu_rng_seed(C.c_int(self.n_chain), C.c_uint64(int(seed)), C.c_uint64(int(first_stream)), _p(self.rng))
        L.bfemu_chain_init(C.c_int(self.n_chain), C.c_int(self.d), _p(x_0), C.c_double(step_size), None, None,
                           C.c_double(initial_weight), C.c_int(adapt_window), _p(self.sc), _p(self.vec))
        self.i_iter = 0

    def run(self, n_run, sampler='NUTS', n_warmup=500, max_treedepth=10, n_int_step=32, max_change=1000.,
            target_accept=0.8, gamma=0.05, k=0.75, t_0=10., adapt_step_size=True, adapt_metric=True, update_window=1,
            doubling=True, launch_iters=None, layout='group'):
        bl = self.bl
        cfg = bl.SamplerConfig()
        cfg.sampler = {'NUTS': 0, 'HMC': 1}[sampler]
        cfg.n_warmup = int(n_warmup)
        cfg.max_treedepth = int(max_treedepth)
        cfg.n_int_step = int(n_int_step)
        cfg.max_change = float(max_change)
        cfg.target_accept, cfg.gamma, cfg.k, cfg.t_0 = float(target_accept), float(gamma), float(k), float(t_0)
        cfg.adapt_step_size, cfg.adapt_metric = int(bool(adapt_step_size)), int(bool(adapt_metric))
        cfg.update_window, cfg.doubling = int(update_window), int(bool(doubling))
        cfg.chain_layout = {'group': 1, 'split': 3}[layout]
        samples = np.full((self.n_chain, n_run, self.d), np.nan)
        stats = np.full((self.n_chain, n_run, bl.STAT_STRIDE), np.nan)
        step = max(1, int(launch_iters) if launch_iters else n_run)
        f = lib().bfemu_sampler_run
        f.restype = C.c_int
        for done in range(step, n_run + step, step):
            rc = f(C.byref(self.ds), C.byref(cfg), C.c_int(self.n_chain), C.c_int(self.i_iter + min(done, n_run)),
                   _p(self.rng), _p(self.sc), _p(self.vec), C.c_int(self.i_iter), C.c_int(n_run), _p(samples), _p(stats),
                   _p(self.n_leapfrog))
            if rc != 0:
                raise RuntimeError('bfemu_sampler_run failed: %d' % rc)
        self.i_iter += n_run
        names = bl.NSTATS if sampler == 'NUTS' else bl.HSTATS
        return samples, {k: stats[:, :, i] for i, k in enumerate(names)}

    def field(self, name):
        bl = self.bl
        if name in bl.SC_FIELDS:
            return self.sc[:, bl.SC_FIELDS.index(name)]
        return self.vec[:, bl.VEC_FIELDS.index(name)]
