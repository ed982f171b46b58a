"""The 64-d ring of the reference's examples/ring-gbs.ipynb (cell 4): logp = -sum_i (x_i^2 + x_{i+1}^2 - a)^2 / b (cyclic) under
a flat prior on [-5, 5]^D, a = 2, b = 1; fiducial logZ = -114.492 (the notebook's last cell; its own run printed -114.473 +-
0.065).  There is no exact sampler for it, so the evidence test draws its posterior samples with a plain vectorised HMC
(NumPy, many chains at once) -- test infrastructure, independent of the package."""
import numpy as np

D, A, B = 64, 2., 1.
CONST = D * np.log(10.)


def logp(x):
    x2 = np.asarray(x, dtype=np.float64)**2
    return -np.sum((x2 + np.roll(x2, -1, axis=-1) - A)**2, axis=-1) / B - CONST


def grad(x):
    x = np.asarray(x, dtype=np.float64)
    x2 = x * x
    t = (x2 + np.roll(x2, -1, axis=-1) - A) / B          # term i couples x_i and x_{i+1}
    return -(2 * t + 2 * np.roll(t, 1, axis=-1)) * 2 * x


def hmc_draws(n_chain=8, n_keep=1500, seed=0, n_parallel=240, n_burn=400, eps=0.06, n_leap=24, thin=3):
    """(n_chain, n_keep, D) posterior draws: n_parallel independent HMC chains (jittered step size, unit metric), started on the
    ring with random signs, n_burn burn-in transitions, then every thin-th state; the chains' draws are dealt into n_chain rows."""
    rng = np.random.default_rng(seed)
    need = -(-n_chain * n_keep // n_parallel)
    x = rng.choice([-1., 1.], size=(n_parallel, D)) + 0.05 * rng.normal(size=(n_parallel, D))
    lp = logp(x)
    out, acc = [], []
    for it in range(n_burn + need * thin):
        p = rng.normal(size=x.shape)
        e = eps * rng.uniform(0.7, 1.3, size=(n_parallel, 1))
        xn, pn = x.copy(), p + 0.5 * e * grad(x)
        for l in range(n_leap):
            xn = xn + e * pn
            pn = pn + (e if l < n_leap - 1 else 0.5 * e) * grad(xn)
        lpn = logp(xn)
        ok = np.log(rng.uniform(size=n_parallel)) < (lpn - 0.5 * np.sum(pn * pn, -1)) - (lp - 0.5 * np.sum(p * p, -1))
        ok &= np.all(np.abs(xn) < 5., axis=-1)
        x = np.where(ok[:, None], xn, x)
        lp = np.where(ok, lpn, lp)
        acc.append(ok.mean())
        if it >= n_burn and (it - n_burn) % thin == thin - 1:
            out.append(x.copy())
    s = np.stack(out, 1).reshape(-1, D)                  # (n_parallel * need, D)
    rng.shuffle(s)
    return s[:n_chain * n_keep].reshape(n_chain, n_keep, D), float(np.mean(acc[n_burn:]))
