from .sample_trace import NTrace, HTrace, TNTrace, GaussianBase, TraceTuple, ChainView, _get_step_size, _get_metric

__all__ = ['NTrace', 'HTrace', 'TNTrace', 'GaussianBase', 'TraceTuple', 'ChainView', '_get_step_size', '_get_metric']
