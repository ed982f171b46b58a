"""Robustness sweep: odd dimensions and chain counts, NUTS and HMC, device vs oracle (discrete fields + head positions)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
from oracle import oracle as orc
ctx = get_context(0)
bad = 0
for d in (1, 2, 3, 15, 17, 33, 48, 65, 100, 128):
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    for C in (1, 15, 17, 100):
        x0 = np.random.default_rng(d * 1000 + C).normal(size=(C, d)) * 0.7
        for smp in ('NUTS', 'HMC'):
            dc = DeviceChains(dens, x0, seed=9)
            s, st = dc.run(14, smp, n_warmup=9, n_int_step=6)
            s, st = s.cpu().numpy(), st.cpu().numpy()
            for i in sorted(set((0, C // 2, C - 1))):
                ch = orc.Chain(x0[i])
                rng = orc.make_rng('xoshiro', seed=9, stream=i)
                if smp == 'NUTS':
                    so, sto = orc.nuts_run(spec, ch, rng, 14, 9)
                    ok = np.array_equal(st[i, :, _lib.NSTATS.index('tree_size')], sto['tree_size'])
                else:
                    so, sto = orc.hmc_run(spec, ch, rng, 14, 9, n_int_step=6)
                    ok = np.array_equal(st[i, :, _lib.HSTATS.index('accepted')], sto['accepted'])
                err = np.abs(s[i, :5] - so[:5]).max()
                if not ok or not err < 1e-8:
                    bad += 1
                    print('MISMATCH d=%d C=%d %s chain %d: discrete ok %s, head err %.2e' % (d, C, smp, i, ok, err))
print('sweep done, mismatches:', bad)
