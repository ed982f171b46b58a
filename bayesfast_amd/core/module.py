"""Surrogate base class with the reference's constructor and wrapper semantics (bayesfast/core/module.py:558-687
and the ModuleBase wrappers :120-227 it inherits), reduced to what the surrogate path needs."""
from collections import namedtuple

import numpy as np

__all__ = ['Surrogate', 'SurrogateScope']

SurrogateScope = namedtuple('SurrogateScope', ['i_step', 'n_step'])


class Surrogate:
    """Base class for surrogate modules.

    input_size, output_size : positive int
    scope : (i_step, n_step), the modules of the true pipeline this surrogate replaces (core/module.py:569-573)
    input_scales : None or (input_size, 2) array; inputs are mapped to (x - lo) / (hi - lo) before ``_fun`` and the
        Jacobian is divided by (hi - lo) afterwards (core/module.py:76-83,226)
    """

    def __init__(self, input_size=None, output_size=None, scope=(0, 1), fit_options=None, input_vars='__var__',
                 output_vars='__var__', delete_vars=(), input_scales=None, label=None):
        try:
            self._input_size = int(input_size)
            assert self._input_size > 0
        except Exception:
            raise ValueError('input_size should be a positive int.')
        try:
            self._output_size = int(output_size)
            assert self._output_size > 0
        except Exception:
            raise ValueError('output_size should be a positive int.')
        self.scope = scope
        self.fit_options = fit_options
        self.input_vars = [input_vars] if isinstance(input_vars, str) else list(input_vars)
        self.output_vars = [output_vars] if isinstance(output_vars, str) else list(output_vars)
        self._delete_vars = [delete_vars] if isinstance(delete_vars, str) else list(delete_vars)
        self.input_scales = input_scales
        self.label = label
        self.reset_counter()

    input_size = property(lambda self: self._input_size)
    output_size = property(lambda self: self._output_size)

    @property
    def scope(self):
        return self._scope

    @scope.setter
    def scope(self, s):
        try:
            i_step, n_step = s
            assert n_step > 0
            self._scope = SurrogateScope(int(i_step), int(n_step))
        except Exception:
            raise ValueError('invalid value for scope.')

    @property
    def fit_options(self):
        return self._fit_options

    @fit_options.setter
    def fit_options(self, options):
        self._fit_options = {} if options is None else dict(options)

    @property
    def input_scales(self):
        return self._input_scales

    @input_scales.setter
    def input_scales(self, scales):
        if scales is None:
            self._input_scales = None
            self._input_scales_diff = 1.
        else:
            try:
                scales = np.ascontiguousarray(scales, dtype=np.float64)
                if scales.ndim == 1:
                    scales = np.array((np.zeros_like(scales), scales)).T.copy()
                assert scales.ndim == 2 and scales.shape == (self._input_size, 2)
                assert np.all(scales[:, 1] > scales[:, 0])
            except Exception:
                raise ValueError('invalid value for input_scales.')
            self._input_scales = scales
            self._input_scales_diff = scales[:, 1] - scales[:, 0]

    def reset_counter(self):
        self._ncall_fun = self._ncall_jac = self._ncall_fun_and_jac = 0

    ncall_fun = property(lambda self: self._ncall_fun)
    ncall_jac = property(lambda self: self._ncall_jac)
    ncall_fun_and_jac = property(lambda self: self._ncall_fun_and_jac)

    def _scale_in(self, args):
        x = np.concatenate([np.atleast_1d(np.asarray(a, dtype=np.float64)) for a in args])
        if self._input_scales is not None:
            x = (x - self._input_scales[:, 0]) / self._input_scales_diff
        return x

    @property
    def fun(self):
        self._ncall_fun += 1
        return lambda *args: [self._fun(self._scale_in(args))]

    __call__ = fun

    @property
    def jac(self):
        self._ncall_jac += 1
        return lambda *args: [self._jac(self._scale_in(args)) / self._input_scales_diff]

    @property
    def fun_and_jac(self):
        self._ncall_fun_and_jac += 1

        def _faj(*args):
            f, j = self._fun_and_jac(self._scale_in(args))
            return [f], [j / self._input_scales_diff]
        return _faj

    def fit(self, *args, **kwargs):
        raise NotImplementedError('Abstract Method.')

    @property
    def n_param(self):
        raise NotImplementedError('Abstract Property.')
