from .module import Surrogate

__all__ = ['Surrogate']
