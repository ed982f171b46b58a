"""Refit glue either side of the sampler inside ``Recipe._sam_step`` / ``_pos_step``:
choosing the points at which the true model is evaluated for the next surrogate fit, with the ``logp_cutoff``
filter (bayesfast/core/recipe.py:1060-1155), and the truncated importance weights of the post-processing step
(core/recipe.py:1270-1297).  Host-side control logic around user callables, as in the reference."""
import warnings

import numpy as np

from ..utils.resample import SystematicResampler

__all__ = ['select_fit_points', 'importance_weights']


def select_fit_points(prev_samples, prev_logq, logp_true, n_eval, resampler=None, logp_cutoff=True, alpha_min=0.75,
                      alpha_supp=1.25):
    """Points and true log-densities for the next ``Density.fit``.

    prev_samples (N, d), prev_logq (N,): the previous round's samples (original space) and the surrogate log-density
    they were drawn from; ``logp_true(x (k, d)) -> (k,)`` evaluates the true model; ``n_eval`` points are picked by
    ``resampler(prev_logq, n_eval)`` (``SystematicResampler`` by default, core/recipe.py:1074-1075).

    With ``logp_cutoff`` (recipe.py:1097-1155) points whose true logp falls below the smallest logq among the
    resampled points are dropped, and supplementary points are drawn until ``alpha_min * n_eval`` good points
    remain (``alpha_supp`` oversamples each supplement).  Returns ``(x_fit, logp_fit, n_true_evaluations)``."""
    prev_samples = np.asarray(prev_samples, dtype=np.float64)
    prev_logq = np.asarray(prev_logq, dtype=np.float64).reshape(-1)
    if prev_samples.ndim != 2 or prev_samples.shape[0] != prev_logq.size:
        raise ValueError('prev_samples (N, d) and prev_logq (N,) do not match.')
    n_eval = int(n_eval)
    if n_eval <= 0:
        raise ValueError('n_eval should be a positive int.')
    if prev_samples.shape[0] < n_eval:  # recipe.py:1060-1065
        raise RuntimeError('I need {} points to fit the surrogate model, but I can find at most {} points in the '
                           'previous step.'.format(n_eval, prev_samples.shape[0]))
    if resampler is None:
        resampler = SystematicResampler()
    i_resample = resampler(prev_logq, n_eval)
    x_fit = prev_samples[i_resample]
    logp_fit = np.asarray(logp_true(x_fit), dtype=np.float64).reshape(-1)
    n_calls = x_fit.shape[0]
    if not logp_cutoff:
        return x_fit, logp_fit, n_calls
    logq_min = np.min(prev_logq[i_resample])
    is_good = logp_fit > logq_min
    f_good = np.sum(is_good) / logp_fit.size
    if f_good < 0.5:
        warnings.warn('more than half of the samples are abandoned because their logp < logq_min.', RuntimeWarning)
    if f_good == 0.:
        raise RuntimeError('f_good is 0, indicating that the samples seem very bad. Please check your recipe setup. '
                           'You may also want to try logp_cutoff=False for the SampleStep.')
    x_fit, logp_fit = x_fit[is_good], logp_fit[is_good]
    n_eval_min = int(alpha_min * n_eval)
    # (the reference calls np.delete without keeping the result, so resampled points stay in the pool; same here)
    while x_fit.shape[0] < n_eval_min:
        n_supp = max(int((n_eval_min - x_fit.shape[0]) / f_good * alpha_supp), 4)
        if prev_samples.shape[0] < n_supp:
            raise RuntimeError('I do not have enough supplementary points.')
        i_supp = resampler(prev_logq, n_supp)
        x_supp = prev_samples[i_supp]
        logp_supp = np.asarray(logp_true(x_supp), dtype=np.float64).reshape(-1)
        n_calls += x_supp.shape[0]
        good = logp_supp > logq_min
        if np.sum(good) < logp_supp.size / 2:
            warnings.warn('more than half of the samples are abandoned because their logp < logq_min.', RuntimeWarning)
        x_fit = np.concatenate((x_fit, x_supp[good]))
        logp_fit = np.concatenate((logp_fit, logp_supp[good]))
    return x_fit, logp_fit, n_calls


def importance_weights(logp, logq, k_trunc=0.25):
    """Truncated importance weights of ``PostStep`` (core/recipe.py:1289-1296): ``w = exp(logp - logq)`` clipped at
    ``mean(w) * n ** k_trunc`` (no clipping for ``k_trunc < 0``).  Returns ``(weights, weights_trunc)``."""
    logp = np.asarray(logp, dtype=np.float64).reshape(-1)
    logq = np.asarray(logq, dtype=np.float64).reshape(-1)
    if logp.shape != logq.shape:
        raise ValueError('logp and logq should have the same size.')
    weights = np.exp(logp - logq)
    if k_trunc < 0:
        return weights, weights.copy()
    return weights, np.clip(weights, 0, np.mean(weights) * logp.size**k_trunc)
