from .resample import SystematicResampler

__all__ = ['SystematicResampler']
