"""(executed = the upper triangle of 64 x 64 blocks, n P (P + 64) flops; the full-product equivalent is what a dgemm would be credited with)
bfhip_gram on its own at the headline size (n = 4290, P = 2145) and at config 5's (n = 18402, P = 9201): ms, TFLOP/s, share
of the FP64 MFMA peak; the result against torch's A^T A."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd import _lib
from bayesfast_amd.device import get_context, _ptr
ctx = get_context(0)
L = ctx._lib
for n, P in ((4290, 2145), (18402, 9201), (1122, 561)):
    A = torch.randn((n, P), dtype=torch.float64, device=ctx.device)
    y = torch.randn((n, 1), dtype=torch.float64, device=ctx.device)
    G, r = ctx.empty((P, P)), ctx.empty((P, 1))
    f = lambda: _lib.check(L.bfhip_gram(ctx.handle, n, P, 1, _ptr(A), P, _ptr(y), _ptr(G), _ptr(r)))
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    reps = 10 if P < 5000 else 3
    e0.record(ctx.stream)
    for _ in range(reps):
        f()
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    ref = A.T @ A
    e0.record(); ref = A.T @ A; e1.record(); torch.cuda.synchronize()
    t_blas = e0.elapsed_time(e1) * 1e-3   # (rocBLAS dgemm, the full product: what the matrix pipe sustains on this box)
    err = float((G - ref).abs().max() / ref.abs().max())
    print(json.dumps({'n': n, 'P': P, 'ms': t * 1e3, 'TFLOPs_full_product_equivalent': 2. * n * P * P / t / 1e12, 'TFLOPs_executed': n * P * (P + 64.) / t / 1e12, 'mfma_frac_executed': n * P * (P + 64.) / t / 78.6e12,
                      'rocblas_full_product_ms': t_blas * 1e3, 'rocblas_TFLOPs': 2. * n * P * P / t_blas / 1e12, 'max_rel_err_vs_torch': err, 'atb_err': float((r - A.T @ y).abs().max())}))
