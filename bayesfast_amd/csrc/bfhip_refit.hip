// bfhip_refit.hip -- refit glue on the device: the rank selection behind SystematicResampler
// (utils/misc.py:61-108: `np.argsort(a)[i_all]`), its sharded form (counts of keys below a query, for the exact
// distributed selection of bayesfast_amd/core/refit.py), and the truncated importance weights of PostStep
// (core/recipe.py:1289-1296).  HBM-bound integer / elementwise work: one pass per kernel, the sort is rocPRIM's
// device radix sort (a plain library primitive, like rocBLAS for a plain GEMM).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "bfhip_common.h"

// float64 -> uint64 whose unsigned order is the numeric order, every NaN last and -0 == +0 (numpy's sort order)
__device__ inline uint64_t bf_order_key(double v) {
    if (v != v) return ~0ull;
    if (v == 0.) v = 0.;
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ void bf_keys_iota_kernel(long n, const double *__restrict__ a, uint64_t *__restrict__ keys, int64_t *__restrict__ idx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        keys[i] = bf_order_key(a[i]);
        idx[i] = i;
    }
}

static int ensure_scratch(bfhip_ctx *ctx, size_t need) {
    if (ctx->scratch_bytes >= need) return 0;
    BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) BF_HIP_CHECK(hipFree(ctx->scratch));
    ctx->scratch = NULL;
    ctx->scratch_bytes = 0;
    BF_HIP_CHECK(hipMalloc(&ctx->scratch, need));
    ctx->scratch_bytes = need;
    return 0;
}

extern "C" int bfhip_sort_keys(bfhip_ctx *ctx, long n, const double *a, uint64_t *keys_sorted, int64_t *order) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || (n > 0 && (!a || !keys_sorted || !order)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_sort_keys: invalid argument");
    if (n == 0) return 0;
    if (n > 0x7fffffffL) return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_sort_keys: more than 2^31-1 elements");
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const int64_t *)nullptr,
                                             (int64_t *)nullptr, (size_t)n, 0, 64, ctx->stream);
    if (e != hipSuccess) return bf_set_error(BFHIP_ERR_HIP, "rocprim::radix_sort_pairs (size query): %s", hipGetErrorString(e));
    // unsorted keys and indices live in the context's workspace, next to rocPRIM's temporary storage
    const size_t kb = ((size_t)n * 8 + 255) / 256 * 256;
    if (int rc = ensure_scratch(ctx, 2 * kb + tmp)) return rc;
    uint64_t *k0 = (uint64_t *)ctx->scratch;
    int64_t *i0 = (int64_t *)((char *)ctx->scratch + kb);
    void *t = (char *)ctx->scratch + 2 * kb;
    hipLaunchKernelGGL(bf_keys_iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, a, k0, i0);
    BF_HIP_CHECK(hipGetLastError());
    e = rocprim::radix_sort_pairs(t, tmp, k0, keys_sorted, i0, order, (size_t)n, 0, 64, ctx->stream);  // stable
    if (e != hipSuccess) return bf_set_error(BFHIP_ERR_HIP, "rocprim::radix_sort_pairs: %s", hipGetErrorString(e));
    return 0;
}

__global__ void bf_order_keys_kernel(long n, const double *__restrict__ a, uint64_t *__restrict__ keys) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = bf_order_key(a[i]);
}

extern "C" int bfhip_order_keys(bfhip_ctx *ctx, long n, const double *a, uint64_t *keys) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || (n > 0 && (!a || !keys))) return bf_set_error(BFHIP_ERR_ARG, "bfhip_order_keys: invalid argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bf_order_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, a, keys);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// counts[i] = #{ k : keys_sorted[k] < q[i] }  (upper = 0)  or  #{ k : keys_sorted[k] <= q[i] }  (upper = 1)
__global__ void bf_count_keys_kernel(long n, const uint64_t *__restrict__ ks, long nq, const uint64_t *__restrict__ q, int upper,
                                     int64_t *__restrict__ counts) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint64_t v = q[i];
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        const uint64_t k = ks[mid];
        if (upper ? (k <= v) : (k < v)) lo = mid + 1;
        else hi = mid;
    }
    counts[i] = lo;
}

extern "C" int bfhip_count_keys(bfhip_ctx *ctx, long n, const uint64_t *keys_sorted, long nq, const uint64_t *q, int upper,
                                int64_t *counts) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || nq < 0 || (n > 0 && !keys_sorted) || (nq > 0 && (!q || !counts)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_count_keys: invalid argument");
    if (nq == 0) return 0;
    hipLaunchKernelGGL(bf_count_keys_kernel, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, ctx->stream, n, keys_sorted, nq, q,
                       upper, counts);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- truncated importance weights: w = exp(logp - logq), w_trunc = clip(w, 0, mean(w) n^k) ---------------------
__global__ void bf_iw_exp_kernel(long n, const double *__restrict__ logp, const double *__restrict__ logq, double *__restrict__ w,
                                 double *__restrict__ partial) {
    __shared__ double red[256];
    double s = 0.;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = exp(logp[i] - logq[i]);
        w[i] = v;
        s += v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void bf_iw_clip_kernel(long n, const double *__restrict__ w, const double *__restrict__ partial, int n_part, double k_trunc,
                                  double *__restrict__ wt) {
    // every block adds the partial sums in the same order: one deterministic mean
    double tot = 0.;
    for (int i = 0; i < n_part; ++i) tot += partial[i];
    const double cap = (tot / (double)n) * pow((double)n, k_trunc);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = w[i];
        wt[i] = k_trunc < 0. ? v : (v < 0. ? 0. : (v > cap ? cap : v));  // np.clip(w, 0, cap) (NaN stays)
    }
}

extern "C" int bfhip_importance_weights(bfhip_ctx *ctx, long n, const double *logp, const double *logq, double k_trunc, double *w,
                                        double *w_trunc) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || (n > 0 && (!logp || !logq || !w || !w_trunc)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_importance_weights: invalid argument");
    if (n == 0) return 0;
    const int nb = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
    if (int rc = ensure_scratch(ctx, 256 * sizeof(double))) return rc;
    double *partial = (double *)ctx->scratch;
    hipLaunchKernelGGL(bf_iw_exp_kernel, dim3(nb), dim3(256), 0, ctx->stream, n, logp, logq, w, partial);
    hipLaunchKernelGGL(bf_iw_clip_kernel, dim3(nb), dim3(256), 0, ctx->stream, n, w, partial, nb, k_trunc, w_trunc);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
