from .bridge import bridge
from .gbs import GBS

__all__ = ['bridge', 'GBS']
