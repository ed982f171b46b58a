"""Host BLAS threads.

The reference runs its host linear algebra with BLAS limited to one thread (core/sample.py:167).  Here it matters for another
reason: OpenBLAS worker threads spin for a while after a multi-threaded call, and on a GPU host that starves the ROCm
runtime's own threads -- a sampling launch right after a fit whose bound statistics used a 64-thread dgemm took 160 ms
instead of 67 (measured, tools/refit_debug.py).  The few host products of this package (4290 x 64 by 64 x 64) gain nothing
from threads.

Round 6, the mechanism: in a container with a CPU quota (cgroup cpu.max, e.g. 16 CPUs of a 256-CPU host) the spinning workers
spend the quota of a 100 ms scheduling period in a few milliseconds, and the kernel then freezes EVERY thread of the process -- the
main thread and the ROCm runtime's submission threads included -- until the period ends.  A config-5 GBS run showed nine GPU-idle
gaps of 20-80 ms, each ending exactly on the 100 ms grid (rocprofv3 kernel trace), 0.5 s of its 1.8 s; with BLAS on one thread they
are gone.  ``sample()``, the fits, ``SIT.fit`` and ``GBS.run`` therefore run inside this context, as the reference's sampler does
(core/sample.py:167)."""
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
