from .poly import PolyConfig, PolyModel

__all__ = ['PolyConfig', 'PolyModel']
