"""Host BLAS threads.

The reference runs its host linear algebra with BLAS limited to one thread (core/sample.py:167).  Here it matters for another
reason: OpenBLAS worker threads spin for a while after a multi-threaded call, and on a GPU host that starves the ROCm
runtime's own threads -- a sampling launch right after a fit whose bound statistics used a 64-thread dgemm took 160 ms
instead of 67 (measured, tools/refit_debug.py).  The few host products of this package (4290 x 64 by 64 x 64) gain nothing
from threads."""
import contextlib

__all__ = ['blas_single_thread']

_controller = None


def blas_single_thread():
    """Context manager: BLAS calls inside run on one thread (a no-op without threadpoolctl)."""
    global _controller
    try:
        if _controller is None:
            from threadpoolctl import ThreadpoolController
            _controller = ThreadpoolController()
        return _controller.limit(limits=1, user_api='blas')
    except Exception:
        return contextlib.nullcontext()
