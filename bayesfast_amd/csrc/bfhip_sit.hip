// bfhip_sit.hip -- data-parallel pieces of the evidence path: Gaussianized bridge sampling =
// SIT (transforms/sit.py:223-459) + bridge (evidence/bridge.py:10-76).
//
//   bfhip_kde_cdf        1-d weighted Gaussian KDE cdf of every dimension at a set of points (utils/kde.py:322-354);
//                        the Gaussianizing map of SIT._gaussianize_1d is norm.ppf of it at the spline knots
//   bfhip_ndtri          the normal quantile function (scipy.special.ndtri = Cephes) of an array: the Sobol-normal draws of
//                        SIT.sample and the values at the spline knots
//   bfhip_spline_apply   evaluate / derivative / solve of the per-dimension piecewise cubics (utils/_cubic.pyx:188-336)
//                        for a batch of points: SIT.forward_transform / backward_transform / logq
//   bfhip_bridge_sums    the two log-sum-exp terms of the bridge estimator's score function (evidence/bridge.py:44-49)
//   bfhip_bridge_terms   the per-sample terms f1, f2 of its error estimate (:52-57)
//
// HBM-/ALU-bound elementwise and reduction work in float64; every reduction has a fixed order (per-block partial sums
// combined by one thread), so results do not depend on the launch.
#include <cmath>
#include "bfhip_common.h"
#include "bfhip_ndtri.h"

static int ensure_ws(bfhip_ctx *ctx, size_t need) {
    if (ctx->scratch_bytes >= need) return 0;
    BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) BF_HIP_CHECK(hipFree(ctx->scratch));
    ctx->scratch = NULL;
    ctx->scratch_bytes = 0;
    BF_HIP_CHECK(hipMalloc(&ctx->scratch, need));
    ctx->scratch_bytes = need;
    return 0;
}

// ---- KDE cdf -----------------------------------------------------------------------------------------------------
// out[j][i] = sum_k w[k] ndtr((pts[j][i] - data[j][k]) / h[j]);  data (d, n), pts (d, m): one row per dimension.
// grid (point tiles, data splits, d); a thread keeps KP points in registers and strides over its split of the data.
#define KDE_KP 8
#define KDE_TH 256
__global__ __launch_bounds__(KDE_TH) void bf_kde_cdf_kernel(int n, int m, const double *__restrict__ data, const double *__restrict__ w,
                                                           const double *__restrict__ h, const double *__restrict__ pts, int n_split,
                                                           double *__restrict__ partial) {
    const int j = blockIdx.z, sp = blockIdx.y, i0 = blockIdx.x * KDE_KP;
    const double inv = 0.70710678118654752440 / h[j];  // ndtr(z) = erfc(-z / sqrt 2) / 2
    const double *dj = data + (size_t)j * n;
    double p[KDE_KP], acc[KDE_KP];
#pragma unroll
    for (int t = 0; t < KDE_KP; ++t) {
        p[t] = (i0 + t < m) ? pts[(size_t)j * m + i0 + t] : 0.;
        acc[t] = 0.;
    }
    const long per = ((long)n + n_split - 1) / n_split, k0 = (long)sp * per, k1 = (k0 + per < n) ? k0 + per : n;
    for (long k = k0 + threadIdx.x; k < k1; k += KDE_TH) {
        const double xk = dj[k], wk = 0.5 * w[k];
#pragma unroll
        for (int t = 0; t < KDE_KP; ++t) acc[t] += wk * erfc((xk - p[t]) * inv);
    }
    __shared__ double red[KDE_TH];
#pragma unroll
    for (int t = 0; t < KDE_KP; ++t) {
        red[threadIdx.x] = acc[t];
        __syncthreads();
        for (int o = KDE_TH / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0 && i0 + t < m) partial[((size_t)j * n_split + sp) * m + i0 + t] = red[0];
        __syncthreads();
    }
}

__global__ void bf_kde_combine_kernel(int d, int m, int n_split, const double *__restrict__ partial, double *__restrict__ out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)d * m) return;
    const int j = (int)(e / m), i = (int)(e % m);
    double s = 0.;
    for (int sp = 0; sp < n_split; ++sp) s += partial[((size_t)j * n_split + sp) * m + i];
    out[e] = s;
}

extern "C" int bfhip_kde_cdf(bfhip_ctx *ctx, int d, long n, const double *data, const double *w, const double *h, int m,
                             const double *pts, double *out) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || d < 1 || n < 1 || m < 0 || !data || !w || !h || (m > 0 && (!pts || !out)) || n > 0x7fffffffL)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_kde_cdf: invalid argument");
    if (m == 0) return 0;
    const int tiles = (m + KDE_KP - 1) / KDE_KP;
    int n_split = (int)((n + 16383) / 16384);
    const int want = (4 * ctx->n_cu + tiles * d - 1) / (tiles * d);  // enough workgroups to fill the chip
    if (n_split > want) n_split = want;
    if (n_split < 1) n_split = 1;
    if (int rc = ensure_ws(ctx, (size_t)d * n_split * m * sizeof(double))) return rc;
    double *partial = (double *)ctx->scratch;
    hipLaunchKernelGGL(bf_kde_cdf_kernel, dim3(tiles, n_split, d), dim3(KDE_TH), 0, ctx->stream, (int)n, m, data, w, h, pts, n_split,
                       partial);
    hipLaunchKernelGGL(bf_kde_combine_kernel, dim3((unsigned)(((long)d * m + 255) / 256)), dim3(256), 0, ctx->stream, d, m, n_split,
                       partial, out);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- the normal quantile function ------------------------------------------------------------------------------------
// scipy.special.ndtri (what scipy.stats.norm.ppf evaluates; the reference calls it at utils/sobol.py:57 on the Sobol points and at
// transforms/sit.py:225 on the KDE cdfs) is Cephes' ndtri.c (Moshier): a rational function of (p - 1/2)^2 in the middle
// (|p - 1/2| <= 1/2 - exp(-2)), and in the tails, with z = sqrt(-2 log p), z - log(z) / z minus a rational function of 1 / z
// (one for z < 8, one beyond).  Restated here with Cephes' published coefficient tables; scipy itself is a third-party
// dependency of the reference (not under /root/reference), tests/test_evidence.py pins this against scipy's values.
__global__ void bf_ndtri_kernel(long n, const double *__restrict__ p, double *__restrict__ out) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) out[e] = bf_ndtri(p[e]);
}

extern "C" int bfhip_ndtri(bfhip_ctx *ctx, long n, const double *p, double *out) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || (n > 0 && (!p || !out))) return bf_set_error(BFHIP_ERR_ARG, "bfhip_ndtri: invalid argument");
    if (n == 0) return 0;
    long blocks = (n + 255) / 256;
    if (blocks > 16L * ctx->n_cu) blocks = 16L * ctx->n_cu;
    hipLaunchKernelGGL(bf_ndtri_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n, p, out);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- piecewise cubics ----------------------------------------------------------------------------------------------
// Dimension j owns knots x[off[j] .. off[j+1]) (m_j of them), their values y, and m_j + 1 rows of coefficients
// c[(off[j] + j + r) * 4 + 0..3] (highest order first; row 0 and row m_j are the linear extrapolations), exactly the
// arrays of utils/cubic.py:cubic_spline (_x, _y, _c).

// utils/_cubic.pyx:24-96: interval with x[i-1] <= v < x[i]; 0 below x[0], m at or above x[m-1], -1 for NaN
__device__ inline int bf_find_interval(const double *x, int m, double v) {
    if (!(v >= x[0] && v < x[m - 1])) return (v < x[0]) ? 0 : ((v >= x[m - 1]) ? m : -1);
    int lo = 1, hi = m - 1;  // answer in [1, m-1]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v < x[mid]) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}
__device__ inline double bf_cubic_val(const double *c, double x) { return c[0] * x * x * x + c[1] * x * x + c[2] * x + c[3]; }
__device__ inline double bf_cubic_der(const double *c, double x) { return 3 * c[0] * x * x + 2 * c[1] * x + c[2]; }

// mode 0 evaluate (_cubic.pyx:188-231), 1 derivative (:237-279), 2 solve (:285-331, bisection :131-163)
__global__ void bf_spline_apply_kernel(int mode, long n, int d, const double *__restrict__ x, const int *__restrict__ off,
                                       const double *__restrict__ xk, const double *__restrict__ yk, const double *__restrict__ ck,
                                       double *__restrict__ out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    const int j = (int)(e % d);
    const int o = off[j], m = off[j + 1] - o;
    const double *xj = xk + o, *yj = yk + o, *cj = ck + (size_t)(o + j) * 4;
    const double v = x[e];
    const int iv = bf_find_interval(mode == 2 ? yj : xj, m, v);
    double r = NAN;
    if (mode == 0) {
        if (iv > 0 && iv < m) r = bf_cubic_val(cj + 4 * iv, v - xj[iv - 1]);
        else if (iv == 0) r = cj[2] * (v - xj[0]) + cj[3];
        else if (iv == m) r = cj[4 * m + 2] * (v - xj[m - 1]) + cj[4 * m + 3];
    } else if (mode == 1) {
        if (iv > 0 && iv < m) r = bf_cubic_der(cj + 4 * iv, v - xj[iv - 1]);
        else if (iv == 0) r = cj[2];
        else if (iv == m) r = cj[4 * m + 2];
    } else {
        if (iv > 0 && iv < m) {
            const double *c = cj + 4 * iv;
            double a = 0., b = xj[iv] - xj[iv - 1], t = (a + b) / 2, f = bf_cubic_val(c, t) - v;
            int it = 0;
            while (!(f < 1e-10 && f > -1e-10)) {
                if (f > 0) b = t;
                else a = t;
                t = (a + b) / 2;
                f = bf_cubic_val(c, t) - v;
                if (++it >= 100) { t = NAN; break; }
            }
            r = xj[iv - 1] + t;
        } else if (iv == 0) {
            r = xj[0] + (v - cj[3]) / cj[2];
        } else if (iv == m) {
            r = xj[m - 1] + (v - cj[4 * m + 3]) / cj[4 * m + 2];
        }
    }
    out[e] = r;
}

extern "C" int bfhip_spline_apply(bfhip_ctx *ctx, int mode, long n, int d, const double *x, const int *knot_off, const double *knots,
                                  const double *values, const double *coef, double *out) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || mode < 0 || mode > 2 || n < 0 || d < 1 || !knot_off || !knots || !values || !coef || (n > 0 && (!x || !out)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_spline_apply: invalid argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bf_spline_apply_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, ctx->stream, mode, n, d, x, knot_off,
                       knots, values, coef, out);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- bridge sampling ---------------------------------------------------------------------------------------------
// log sum_i exp(t_i),  t_i = (s + a_i) - logaddexp(s + a_i, 0)  (evidence/bridge.py:45-48 with s = logr or -logr)
__device__ inline double bf_log_sigmoid(double x) {  // x - logaddexp(x, 0) = -softplus(-x)
    return (x < 0.) ? x - log1p(exp(x)) : -log1p(exp(-x));
}
__global__ __launch_bounds__(256) void bf_lse_sig_kernel(long n, const double *__restrict__ a, double s, double *__restrict__ part) {
    __shared__ double rm[256], rs[256];
    double mx = -INFINITY, sm = 0.;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double t = bf_log_sigmoid(s + a[i]);
        if (t > mx) { sm = sm * exp(mx - t) + 1.; mx = t; }  // (mx = -inf: exp(-inf) = 0)
        else if (t > -INFINITY) sm += exp(t - mx);
        else if (t != t) sm = NAN;                            // NaN propagates through the sums
    }
    rm[threadIdx.x] = mx;
    rs[threadIdx.x] = sm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double m1 = rm[threadIdx.x], m2 = rm[threadIdx.x + o], s1 = rs[threadIdx.x], s2 = rs[threadIdx.x + o];
            const double mm = m1 > m2 ? m1 : m2;
            rs[threadIdx.x] = (mm == -INFINITY) ? s1 + s2 : s1 * exp(m1 - mm) + s2 * exp(m2 - mm);
            rm[threadIdx.x] = mm;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = rm[0]; part[2 * blockIdx.x + 1] = rs[0]; }
}
__global__ void bf_lse_combine_kernel(int nb_a, int nb_b, const double *__restrict__ part, double *__restrict__ out) {
    if (threadIdx.x >= 2 || blockIdx.x != 0) return;
    const double *p = part + (threadIdx.x == 0 ? 0 : 2 * nb_a);
    const int nb = threadIdx.x == 0 ? nb_a : nb_b;
    double mx = -INFINITY;
    for (int i = 0; i < nb; ++i)
        if (p[2 * i] > mx) mx = p[2 * i];
    double s = 0.;
    for (int i = 0; i < nb; ++i) s += (p[2 * i] == -INFINITY) ? p[2 * i + 1] : p[2 * i + 1] * exp(p[2 * i] - mx);
    out[threadIdx.x] = mx + log(s);
}

extern "C" int bfhip_bridge_sums(bfhip_ctx *ctx, long n_a, const double *a, long n_b, const double *b, double logr, double *out2) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n_a < 1 || n_b < 1 || !a || !b || !out2) return bf_set_error(BFHIP_ERR_ARG, "bfhip_bridge_sums: invalid argument");
    const int nb_a = (int)((n_a + 255) / 256 < 512 ? (n_a + 255) / 256 : 512), nb_b = (int)((n_b + 255) / 256 < 512 ? (n_b + 255) / 256 : 512);
    if (int rc = ensure_ws(ctx, (size_t)2 * (nb_a + nb_b) * sizeof(double))) return rc;
    double *part = (double *)ctx->scratch;
    hipLaunchKernelGGL(bf_lse_sig_kernel, dim3(nb_a), dim3(256), 0, ctx->stream, n_a, a, logr, part);
    hipLaunchKernelGGL(bf_lse_sig_kernel, dim3(nb_b), dim3(256), 0, ctx->stream, n_b, b, -logr, part + 2 * nb_a);
    hipLaunchKernelGGL(bf_lse_combine_kernel, dim3(1), dim3(64), 0, ctx->stream, nb_a, nb_b, part, out2);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// numpy.logaddexp
__device__ inline double bf_lae(double x, double y) {
    if (x == y) return x + 0.6931471805599453094;
    const double t = x - y;
    if (t > 0) return x + log1p(exp(-t));
    if (t <= 0) return y + log1p(exp(t));
    return t;
}
// evidence/bridge.py:52-57: f1 over the q samples, f2 over the p samples
__global__ void bf_bridge_terms_kernel(long n_p, const double *__restrict__ lpp, const double *__restrict__ lqp, long n_q,
                                       const double *__restrict__ lpq, const double *__restrict__ lqq, double logr, double *__restrict__ f1,
                                       double *__restrict__ f2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const double lp = log((double)n_p / (double)(n_p + n_q)), lq = log((double)n_q / (double)(n_p + n_q));
    if (i < n_q) f1[i] = exp(lpq[i] - logr - bf_lae(lpq[i] - logr + lp, lqq[i] + lq));
    if (i < n_p) f2[i] = exp(lqp[i] - bf_lae(lpp[i] - logr + lp, lqp[i] + lq));
}

extern "C" int bfhip_bridge_terms(bfhip_ctx *ctx, long n_p, const double *logp_p, const double *logq_p, long n_q, const double *logp_q,
                                  const double *logq_q, double logr, double *f1, double *f2) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n_p < 1 || n_q < 1 || !logp_p || !logq_p || !logp_q || !logq_q || !f1 || !f2)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_bridge_terms: invalid argument");
    const long nmax = n_p > n_q ? n_p : n_q;
    hipLaunchKernelGGL(bf_bridge_terms_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, ctx->stream, n_p, logp_p, logq_p, n_q,
                       logp_q, logq_q, logr, f1, f2);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- symmetric decorrelation of FastICA (sklearn.decomposition._fastica._sym_decorrelation; transforms/sit.py:235-244) ----------
// W <- (W W^T)^{-1/2} W is the orthogonal polar factor of W; here by the Newton-Schulz iteration X <- 1.5 X - 0.5 (X X^T) X from
// X_0 = W / sqrt(|W|_1 |W|_inf) (singular values in (0, 1]: monotone, finally quadratic convergence to 1) -- two d x d x d products a
// step on the FP64 matrix cores, one wave per 16 x 16 tile of the result (64 waves on 64 CUs at d = 128; rocBLAS takes ~20 us for
// such a product, 32 steps of two were the whole cost of a device-resident FastICA iteration, transforms/ica.py).  No
// eigen-decomposition and no host round trip: the iteration count is fixed by the caller, the residual max |X X^T - I| of the
// result is left on the device for it to look at when it next synchronises.
typedef double bf_d4 __attribute__((ext_vector_type(4)));

// C = beta Cin + alpha A op(B), d x d row-major; op(B) = B^T when TRANSB
template <bool TRANSB>
__global__ __launch_bounds__(64) void bf_small_gemm_kernel(int d, const double *__restrict__ A, const double *__restrict__ B,
                                                          const double *Cin, double *C, double alpha, double beta) {
    const int nt = (d + 15) / 16, ti = blockIdx.x / nt, tj = blockIdx.x % nt, lane = threadIdx.x;
    const int ar = 16 * ti + (lane & 15), bn = 16 * tj + (lane & 15), kk = lane >> 4;
    const int ns = (d + 3) / 4;
    bf_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = bf_d4{0., 0., 0., 0.};
    for (int s0 = 0; s0 < ns; s0 += 8) {   // eight k-steps of operands on their way together, four accumulation chains
        double a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 4 * (s0 + q) + kk;
            const bool ok = k < d && s0 + q < ns;
            a[q] = (ok && ar < d) ? A[(size_t)ar * d + k] : 0.;
            b[q] = (ok && bn < d) ? (TRANSB ? B[(size_t)bn * d + k] : B[(size_t)k * d + bn]) : 0.;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q & 3], 0, 0, 0);
    }
    const bf_d4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + 4 * r + kk, col = bn;
        if (row < d && col < d) {
            const size_t o = (size_t)row * d + col;
            C[o] = (beta != 0. ? beta * Cin[o] : 0.) + alpha * t[r];
        }
    }
}

// X0 = A / sqrt(|A|_1 |A|_inf) (one workgroup; d <= 1024)
__global__ __launch_bounds__(256) void bf_ns_scale_kernel(int d, const double *__restrict__ A, double *__restrict__ X) {
    __shared__ double red[256];
    __shared__ double s_col, s_row;
    double mc = 0., mr = 0.;
    for (int j = threadIdx.x; j < d; j += 256) {
        double c = 0., r = 0.;
        for (int i = 0; i < d; ++i) { c += fabs(A[(size_t)i * d + j]); r += fabs(A[(size_t)j * d + i]); }
        mc = c > mc ? c : mc;
        mr = r > mr ? r : mr;
    }
    red[threadIdx.x] = mc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) s_col = red[0];
    __syncthreads();
    red[threadIdx.x] = mr;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) s_row = red[0];
    __syncthreads();
    const double inv = 1. / sqrt(s_col * s_row);
    for (int i = threadIdx.x; i < d * d; i += 256) X[i] = A[i] * inv;
}

// resid[0] = max |T - I| over the d x d matrix T (one workgroup)
__global__ __launch_bounds__(256) void bf_ns_resid_kernel(int d, const double *__restrict__ T, double *__restrict__ resid) {
    __shared__ double red[256];
    double m = 0.;
    for (int i = threadIdx.x; i < d * d; i += 256) {
        const double v = fabs(T[i] - ((i / d == i % d) ? 1. : 0.));
        m = (v > m || v != v) ? v : m;   // (NaN propagates)
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { const double u = red[threadIdx.x + o]; if (u > red[threadIdx.x] || u != u) red[threadIdx.x] = u; }
        __syncthreads();
    }
    if (threadIdx.x == 0) resid[0] = red[0];
}

// The whole iteration as ONE launch: nt x nt single-wave workgroups (all resident: d <= 512), a grid barrier between the two
// products of a step (an arrival counter in global memory, targets that only grow: nothing is reset inside the launch).  65 small
// launches per decorrelation were 0.46 ms of host time each call, more than the device needed for them.
__device__ inline void bf_grid_barrier(unsigned int *counter, unsigned int target) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __threadfence();
}

// one 16 x 16 tile of C = beta Cin + alpha A op(B) (the body of bf_small_gemm_kernel)
template <bool TRANSB>
__device__ inline void bf_small_gemm_tile(int d, int ti, int tj, int lane, const double *A, const double *B, const double *Cin, double *C,
                                          double alpha, double beta) {
    const int ar = 16 * ti + (lane & 15), bn = 16 * tj + (lane & 15), kk = lane >> 4;
    const int ns = (d + 3) / 4;
    bf_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = bf_d4{0., 0., 0., 0.};
    for (int s0 = 0; s0 < ns; s0 += 32) {   // 32 k-steps (all of them at d <= 128) of operands on their way together: one L2 latency a product
        double a[32], b[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int k = 4 * (s0 + q) + kk;
            const bool ok = k < d && s0 + q < ns;
            a[q] = (ok && ar < d) ? A[(size_t)ar * d + k] : 0.;
            b[q] = (ok && bn < d) ? (TRANSB ? B[(size_t)bn * d + k] : B[(size_t)k * d + bn]) : 0.;
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q & 3], 0, 0, 0);
    }
    const bf_d4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + 4 * r + kk, col = bn;
        if (row < d && col < d) {
            const size_t o = (size_t)row * d + col;
            C[o] = (beta != 0. ? beta * Cin[o] : 0.) + alpha * t[r];
        }
    }
}

// wave maximum of a non-negative value (NaN wins)
__device__ inline double bf_wave_max_nn(double v) {
    for (int o = 32; o > 0; o >>= 1) { const double u = __shfl_xor(v, o, 64); v = (u > v || u != u) ? u : v; }
    return v;
}

__global__ __launch_bounds__(256) void bf_polar_ns_kernel(int d, const double *a, double *x, int n_iter, double *work, double *resid,
                                                         unsigned int *counter, unsigned long long *dev_slots) {
    // four waves per workgroup, a 16 x 16 tile per wave (16 workgroups at d = 128: fewer arrivals per grid barrier than 64)
    const int nt = (d + 15) / 16, tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool has = tile < nt * nt;
    const int ti = has ? tile / nt : 0, tj = has ? tile % nt : 0;
    const unsigned int nwg = gridDim.x;
    double *T = work, *Y = work + (size_t)d * d;
    double *cur = x;
    unsigned int phase = 0;
    // X_0 = A / sqrt(|A|_1 |A|_inf): every workgroup takes the two norms (d^2 reads from L2, eight rows of loads in flight), then
    // scales its own tile
    {
        double mc = 0., mr = 0.;
        for (int j = lane; j < d; j += 64) {
            double c = 0., r = 0.;
            int i = 0;
            for (; i + 8 <= d; i += 8) {
                double cv[8], rv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { cv[q] = a[(size_t)(i + q) * d + j]; rv[q] = a[(size_t)j * d + i + q]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) { c += fabs(cv[q]); r += fabs(rv[q]); }
            }
            for (; i < d; ++i) { c += fabs(a[(size_t)i * d + j]); r += fabs(a[(size_t)j * d + i]); }
            mc = c > mc ? c : mc;
            mr = r > mr ? r : mr;
        }
        for (int o = 32; o > 0; o >>= 1) { mc = fmax(mc, __shfl_xor(mc, o, 64)); mr = fmax(mr, __shfl_xor(mr, o, 64)); }
        const double inv = 1. / sqrt(mc * mr);
        for (int e = lane; e < 256; e += 64) {
            const int row = 16 * ti + (e >> 4), col = 16 * tj + (e & 15);
            if (has && row < d && col < d) cur[(size_t)row * d + col] = a[(size_t)row * d + col] * inv;
        }
    }
    bf_grid_barrier(counter, ++phase * nwg);
    // The steps.  T = X X^T of a step is also the measure of how far X is from orthogonal: every tile posts max |T - I| (an atomic
    // maximum of the bit pattern of a non-negative double), and once the whole matrix is below 1e-13 the iteration STOPS -- the
    // count follows the matrix (15-25 steps for the noise-dominated updates of a nearly Gaussian SIT iteration, fewer otherwise)
    // instead of a worst-case constant; the same data give the same count.
    double r_last = 1.;
    for (int it = 0; it <= n_iter; ++it) {
        double dv = 0.;
        if (has) {
            bf_small_gemm_tile<true>(d, ti, tj, lane, cur, cur, nullptr, T, 1., 0.);
            // (this wave's own tile of T: the values it just stored)
            const int kk = lane >> 4, col = 16 * tj + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + 4 * r + kk;
                if (row < d && col < d) {
                    const double v = fabs(T[(size_t)row * d + col] - (row == col ? 1. : 0.));
                    dv = (v > dv || v != v) ? v : dv;
                }
            }
            dv = bf_wave_max_nn(dv);
            if (lane == 0) atomicMax(&dev_slots[it], (unsigned long long)__double_as_longlong(dv != dv ? __builtin_inf() : dv));
        }
        bf_grid_barrier(counter, ++phase * nwg);
        r_last = __longlong_as_double((long long)__hip_atomic_load(&dev_slots[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (r_last < 1e-13 || it == n_iter) break;
        double *nxt = (cur == x) ? Y : x;
        if (has) bf_small_gemm_tile<false>(d, ti, tj, lane, T, cur, cur, nxt, -0.5, 1.5);
        bf_grid_barrier(counter, ++phase * nwg);
        cur = nxt;
    }
    if (cur != x && has) {   // the result belongs in x: this wave's tile
        for (int e = lane; e < 256; e += 64) {
            const int row = 16 * ti + (e >> 4), col = 16 * tj + (e & 15);
            if (row < d && col < d) x[(size_t)row * d + col] = cur[(size_t)row * d + col];
        }
    }
    if (tile == 0 && lane == 0) resid[0] = r_last;
}

// Round 6b: ONE grid barrier per step.  A workgroup owns a 16-row block of X (one wave per 16 x 16 tile of it, d <= 256): its waves
// build the block's rows of T = X X^T side by side into LDS (they need all of X, which the barrier of the previous step made
// visible), a workgroup barrier, and the update X' = 1.5 X - 0.5 T X of the block takes T from LDS: T never travels, and the step
// ends in one arrival at the grid barrier (release fence before, acquire fence after: one cache write-back and one invalidate per
// step where the first form paid two of each twice).  X_0 = A / sqrt(|A|_1 |A|_inf) is applied on load in the first step.  The
// arithmetic per tile is the first form's (same operands, same four accumulation chains): the iterates are bit-identical.
__device__ inline void bf_grid_barrier_ra(unsigned int *counter, unsigned int target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// this wave's tile of T = (sc X)(sc X)^T into the workgroup's LDS rows; returns max |T - I| over the tile (NaN -> inf)
template <int KB>
__device__ inline double bf_polar_rows_t(int d, int ti, int tj, int lane, const double *X, double sc, double *Tl, int ld) {
    const int ar = 16 * ti + (lane & 15), bn = 16 * tj + (lane & 15), kk = lane >> 4;
    const int ns = (d + 3) / 4;
    const int aoff = (ar < d ? ar : 0) * d + kk, boff = (bn < d ? bn : 0) * d + kk;
    bf_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = bf_d4{0., 0., 0., 0.};
    for (int s0 = 0; s0 < ns; s0 += KB) {
        double a[KB], b[KB];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            // (a uniform base per k-step plus one per-lane offset: an address register pair per operand stream, not per load)
            const int k = 4 * (s0 + q) + kk;
            const bool ok = k < d && s0 + q < ns;
            const double *col = X + 4 * (s0 + q);
            a[q] = (ok && ar < d) ? col[aoff] * sc : 0.;
            b[q] = (ok && bn < d) ? col[boff] * sc : 0.;
        }
#pragma unroll
        for (int q = 0; q < KB; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q & 3], 0, 0, 0);
    }
    const bf_d4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    double dv = 0.;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lr = 4 * r + kk, row = 16 * ti + lr, col = bn;
        const bool in = row < d && col < d;
        const double tv = in ? 0. + 1. * t[r] : 0.;     // (the first form stored 0 + 1 * t: the same value)
        Tl[lr * ld + col] = tv;
        if (in) {
            const double v = fabs(tv - (row == col ? 1. : 0.));
            dv = (v > dv || v != v) ? v : dv;
        }
    }
    return dv;
}

// this wave's tile of X' = 1.5 (sc X) - 0.5 T (sc X), T's rows from LDS
template <int KB>
__device__ inline void bf_polar_rows_x(int d, int ti, int tj, int lane, const double *X, double sc, const double *Tl, int ld, double *out) {
    const int bn = 16 * tj + (lane & 15), kk = lane >> 4;
    const int ns = (d + 3) / 4;
    const int boff = kk * d + (bn < d ? bn : 0);
    bf_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = bf_d4{0., 0., 0., 0.};
    for (int s0 = 0; s0 < ns; s0 += KB) {
        double a[KB], b[KB];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const int k = 4 * (s0 + q) + kk;
            const bool ok = k < d && s0 + q < ns;
            const double *row = X + (size_t)(4 * (s0 + q)) * d;
            a[q] = ok ? Tl[(lane & 15) * ld + k] : 0.;
            b[q] = (ok && bn < d) ? row[boff] * sc : 0.;
        }
#pragma unroll
        for (int q = 0; q < KB; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q & 3], 0, 0, 0);
    }
    const bf_d4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + 4 * r + kk, col = bn;
        if (row < d && col < d) {
            const size_t o = (size_t)row * d + col;
            out[o] = 1.5 * (X[o] * sc) + -0.5 * t[r];
        }
    }
}

// NW waves per workgroup (= tiles per row block); KB k-steps of operands on their way together: all of a product at d <= 128 with
// the 256 registers eight waves leave a lane, a quarter of one with sixteen waves
template <int NW, int KB>
__global__ __launch_bounds__(64 * NW) void bf_polar_rows_kernel(int d, const double *a, double *x, int n_iter, double *work, double *resid,
                                                            unsigned int *counter, unsigned long long *dev_slots, unsigned long long *stamps) {
    extern __shared__ double bf_polar_tl[];
    int n_stamp = 0;
#define BF_POLAR_STAMP() do { if (stamps && blockIdx.x == 0 && threadIdx.x == 0 && n_stamp < 60) stamps[n_stamp++] = wall_clock64(); } while (0)
    BF_POLAR_STAMP();
    const int nt = (d + 15) / 16, ti = blockIdx.x, tj = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = 16 * nt + 4;
    const unsigned int nwg = gridDim.x;
    double *Y = work;
    // |A|_1 and |A|_inf: every wave for itself (d^2 reads from L2, eight rows of loads in flight)
    double mc = 0., mr = 0.;
    for (int j = lane; j < d; j += 64) {
        double c = 0., r = 0.;
        int i = 0;
        for (; i + 8 <= d; i += 8) {
            double cv[8], rv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { cv[q] = a[(size_t)(i + q) * d + j]; rv[q] = a[(size_t)j * d + i + q]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) { c += fabs(cv[q]); r += fabs(rv[q]); }
        }
        for (; i < d; ++i) { c += fabs(a[(size_t)i * d + j]); r += fabs(a[(size_t)j * d + i]); }
        mc = c > mc ? c : mc;
        mr = r > mr ? r : mr;
    }
    for (int o = 32; o > 0; o >>= 1) { mc = fmax(mc, __shfl_xor(mc, o, 64)); mr = fmax(mr, __shfl_xor(mr, o, 64)); }
    BF_POLAR_STAMP();
    const double *src = a;
    double sc = 1. / sqrt(mc * mr);
    double *dst = x;
    unsigned int phase = 0;
    double r_last = 1.;
    bool converged = false;
    for (int it = 0; it < n_iter; ++it) {
        double dv = bf_polar_rows_t<KB>(d, ti, tj, lane, src, sc, bf_polar_tl, ld);
        dv = bf_wave_max_nn(dv);
        if (lane == 0) atomicMax(&dev_slots[it], (unsigned long long)__double_as_longlong(dv != dv ? __builtin_inf() : dv));
        BF_POLAR_STAMP();
        __syncthreads();
        BF_POLAR_STAMP();
        bf_polar_rows_x<KB>(d, ti, tj, lane, src, sc, bf_polar_tl, ld, dst);
        BF_POLAR_STAMP();
        bf_grid_barrier_ra(counter, ++phase * nwg);
        BF_POLAR_STAMP();
        src = dst;
        sc = 1.;
        dst = (dst == x) ? Y : x;
        // (the measure of the iterate this step STARTED from: below the threshold the step just taken only polished it)
        r_last = __longlong_as_double((long long)__hip_atomic_load(&dev_slots[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (r_last < 1e-13) { converged = true; break; }
    }
    if (!converged) {   // the steps ran out (or there were none): the measure of the iterate they ended on
        double dv = bf_polar_rows_t<KB>(d, ti, tj, lane, src, sc, bf_polar_tl, ld);
        dv = bf_wave_max_nn(dv);
        if (lane == 0) atomicMax(&dev_slots[n_iter], (unsigned long long)__double_as_longlong(dv != dv ? __builtin_inf() : dv));
        bf_grid_barrier_ra(counter, ++phase * nwg);
        r_last = __longlong_as_double((long long)__hip_atomic_load(&dev_slots[n_iter], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (src != x) {   // the result belongs in x: this wave's tile
        for (int e = lane; e < 256; e += 64) {
            const int row = 16 * ti + (e >> 4), col = 16 * tj + (e & 15);
            if (row < d && col < d) x[(size_t)row * d + col] = src[(size_t)row * d + col] * sc;
        }
    }
    if (ti == 0 && tj == 0 && lane == 0) resid[0] = r_last;
}

// Round 6c, d <= 128: the whole of X in LDS.  The stamps of the row-block form said where a step goes: 19 us in the T tiles and
// 7 us in the update against 3 us in the grid barrier -- the operand loads, 16 (or 4) cache lines per instruction straight after the
// barrier's invalidate.  Here every workgroup copies X (<= 128 KB) into its LDS with row-contiguous loads, all of them on their way
// together, and both products take their operands from LDS (row stride 16 nt + 4 doubles: two lanes a bank, the minimum for 64
// eight-byte reads).  |A|_1, |A|_inf and the scaling run on the LDS copy as well.  Same tiles, same accumulation chains: iterates
// bit-identical to the other forms'.
#define BF_POLAR_LDS_MAXT 8
__device__ inline void bf_polar_lds_load(int d, int ld, const double *__restrict__ X, double *Xl, int nthr) {
    // element e = tid + nthr p of the row-major d x d matrix: (row, col) advanced without a division per element
    const int total = d * d, step_r = nthr / d, step_c = nthr % d;
    int e = threadIdx.x, row = e / d, col = e % d;
    for (; e < total;) {
        double v[16];
        int rr[16], cc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            rr[q] = row, cc[q] = col;
            v[q] = (e < total) ? X[e] : 0.;
            e += nthr;
            row += step_r, col += step_c;
            if (col >= d) { col -= d; ++row; }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (rr[q] < d) Xl[rr[q] * ld + cc[q]] = v[q];
    }
}

template <bool TRANSB>
__device__ inline bf_d4 bf_polar_lds_tile(int ns, const double *ap, const double *bp, int ld) {
    // a[s] = ap[4 s] (row fixed per lane, k = 4 s + kk folded into ap); b[s] = TRANSB ? bp[4 s] : bp[4 s ld]
    bf_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = bf_d4{0., 0., 0., 0.};
    for (int s0 = 0; s0 < ns; s0 += 16) {
        double a[16], b[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool ok = s0 + q < ns;
            a[q] = ok ? ap[4 * (s0 + q)] : 0.;
            b[q] = ok ? (TRANSB ? bp[4 * (s0 + q)] : bp[(size_t)4 * (s0 + q) * ld]) : 0.;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc[q & 3], 0, 0, 0);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

__global__ __launch_bounds__(64 * BF_POLAR_LDS_MAXT) void bf_polar_lds_kernel(int d, const double *a, double *x, int n_iter, double *work,
                                                                             double *resid, unsigned int *counter,
                                                                             unsigned long long *dev_slots, unsigned long long *stamps) {
    extern __shared__ double bf_polar_sm[];
    int n_stamp = 0;
    BF_POLAR_STAMP();
    const int nt = (d + 15) / 16, ti = blockIdx.x, tj = threadIdx.x >> 6, lane = threadIdx.x & 63, nthr = blockDim.x;
    const int ld = 16 * nt + 4, kk = lane >> 4;
    const unsigned int nwg = gridDim.x;
    double *Xl = bf_polar_sm, *Tl = bf_polar_sm + 16 * nt * ld;
    double *Y = work;
    for (int e = threadIdx.x; e < (16 * nt + 16) * ld; e += nthr) bf_polar_sm[e] = 0.;   // (rows and columns past d stay zero: no guards in the products)
    __syncthreads();
    bf_polar_lds_load(d, ld, a, Xl, nthr);
    __syncthreads();
    // |A|_1 (column sums) and |A|_inf (row sums) on the LDS copy, each sum in index order by one thread (the other forms' order: the
    // same scale to the last bit): threads 0 .. d-1 a column each, d .. 2d-1 a row each (the workgroup has >= 4 d threads)
    {
        const int t = threadIdx.x;
        if (t < 2 * d) {
            double sum = 0.;
            if (t < d) for (int i = 0; i < d; ++i) sum += fabs(Xl[i * ld + t]);
            else for (int i = 0; i < d; ++i) sum += fabs(Xl[(t - d) * ld + i]);
            Tl[t] = sum;
        }
    }
    __syncthreads();
    double sc;
    {
        double mc = 0., mr = 0.;
        for (int j = 0; j < d; ++j) { mc = Tl[j] > mc ? Tl[j] : mc; mr = Tl[d + j] > mr ? Tl[d + j] : mr; }
        sc = 1. / sqrt(mc * mr);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * nt * ld; e += nthr) Xl[e] = Xl[e] * sc;     // X_0 = A / sqrt(|A|_1 |A|_inf)
    __syncthreads();
    BF_POLAR_STAMP();
    const int ns = (d + 3) / 4;
    const double *src = a;
    double *dst = x;
    unsigned int phase = 0;
    double r_last = 1.;
    bool converged = false, loaded = true;
    const double *arow = Xl + (16 * ti + (lane & 15)) * ld + kk, *brow = Xl + (16 * tj + (lane & 15)) * ld + kk;
    const double *trow = Tl + (lane & 15) * ld + kk, *xcol = Xl + kk * ld + 16 * tj + (lane & 15);
    auto t_tile = [&]() -> double {     // this wave's tile of T = X X^T into Tl; max |T - I| over it
        const bf_d4 t = bf_polar_lds_tile<true>(ns, arow, brow, ld);
        double dv = 0.;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lr = 4 * r + kk, row = 16 * ti + lr, col = 16 * tj + (lane & 15);
            const bool in = row < d && col < d;
            const double tv = in ? 0. + 1. * t[r] : 0.;
            Tl[lr * ld + col] = tv;
            if (in) {
                const double v = fabs(tv - (row == col ? 1. : 0.));
                dv = (v > dv || v != v) ? v : dv;
            }
        }
        return bf_wave_max_nn(dv);
    };
    for (int it = 0; it < n_iter; ++it) {
        if (!loaded) {
            bf_polar_lds_load(d, ld, src, Xl, nthr);
            __syncthreads();
        }
        loaded = false;
        BF_POLAR_STAMP();
        const double dv = t_tile();
        if (lane == 0) atomicMax(&dev_slots[it], (unsigned long long)__double_as_longlong(dv != dv ? __builtin_inf() : dv));
        __syncthreads();
        BF_POLAR_STAMP();
        {
            const bf_d4 t = bf_polar_lds_tile<false>(ns, trow, xcol, ld);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + 4 * r + kk, col = 16 * tj + (lane & 15);
                if (row < d && col < d) dst[(size_t)row * d + col] = 1.5 * Xl[row * ld + col] + -0.5 * t[r];
            }
        }
        BF_POLAR_STAMP();
        bf_grid_barrier_ra(counter, ++phase * nwg);
        BF_POLAR_STAMP();
        src = dst;
        dst = (dst == x) ? Y : x;
        r_last = __longlong_as_double((long long)__hip_atomic_load(&dev_slots[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (r_last < 1e-13) { converged = true; break; }
    }
    if (!converged) {   // the steps ran out (or there were none): the measure of the iterate they ended on
        if (!loaded) {
            bf_polar_lds_load(d, ld, src, Xl, nthr);
            __syncthreads();
        }
        const double dv = t_tile();
        if (lane == 0) atomicMax(&dev_slots[n_iter], (unsigned long long)__double_as_longlong(dv != dv ? __builtin_inf() : dv));
        bf_grid_barrier_ra(counter, ++phase * nwg);
        r_last = __longlong_as_double((long long)__hip_atomic_load(&dev_slots[n_iter], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (src != x) {   // the result belongs in x: this wave's tile (from the LDS copy when no step was taken: X_0)
        for (int e = lane; e < 256; e += 64) {
            const int row = 16 * ti + (e >> 4), col = 16 * tj + (e & 15);
            if (row < d && col < d) x[(size_t)row * d + col] = (src == a) ? Xl[row * ld + col] : src[(size_t)row * d + col];
        }
    }
    if (ti == 0 && tj == 0 && lane == 0) resid[0] = r_last;
}

// ---- the glue of a FastICA iteration (scikit-learn's _ica_par with the logcosh contrast, as SIT calls it: transforms/sit.py:235-244) ----
// Between the two products of an iteration (Y = X1 W^T and G^T X1, library GEMMs) and the polar factor sat ~20 small framework
// kernels and five device-to-device copies; as nodes of the chunk's HIP graph the copies alone cost more than the arithmetic.
// Three kernels replace them.

// G = tanh(Y) in place for Y (n_pad, d) and, per block of ICA_RB rows, the column sums of g'(y) = 1 - tanh(y)^2 over the rows < n
// (rows n .. n_pad are zero padding: they stay zero and are not counted).  partial (n_blocks, d).
#define ICA_RB 32     // (625 workgroups at 20 000 rows: 128-row blocks left a third of the chip idle, 42 us against 15)
__global__ __launch_bounds__(256) void bf_ica_tanh_kernel(long n, long n_pad, int d, double *__restrict__ Y, double *__restrict__ partial) {
    __shared__ double red[256];
    const int cw = d < 256 ? d : 256, ry = 256 / cw;          // cw columns side by side, ry row groups
    const int c0 = threadIdx.x % cw, r0 = threadIdx.x / cw;
    const long row0 = (long)blockIdx.x * ICA_RB;
    for (int cb = 0; cb < d; cb += cw) {
        const int c = cb + c0;
        double sum = 0.;
        if (r0 < ry && c < d)
            for (long r = row0 + r0; r < row0 + ICA_RB && r < n_pad; r += ry) {
                const double g = tanh(Y[r * d + c]);
                Y[r * d + c] = g;
                if (r < n) sum += 1. - g * g;
            }
        red[threadIdx.x] = (r0 < ry && c < d) ? sum : 0.;
        __syncthreads();
        if (r0 == 0 && c < d) {
            double t = 0.;
            for (int q = 0; q < ry; ++q) t += red[q * cw + c0];
            partial[(size_t)blockIdx.x * d + c] = t;
        }
        __syncthreads();
    }
}

// A = (sum_b P[b]) / n - gmean[:, None] W with gmean[i] = (sum_blk partial[blk][i]) / n: one workgroup per row i.
__global__ __launch_bounds__(256) void bf_ica_assemble_kernel(int d, int nb, const double *__restrict__ P, long n, int n_blk,
                                                             const double *__restrict__ partial, const double *__restrict__ W,
                                                             double *__restrict__ A, double *__restrict__ meas_k) {
    __shared__ double gsh[4];
    if (meas_k && blockIdx.x == 0 && threadIdx.x == 0) *meas_k = 0.;   // (the slot bf_ica_post_kernel takes its atomic maximum in)
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double g = 0.;
    for (int b = threadIdx.x; b < n_blk; b += 256) g += partial[(size_t)b * d + i];
    for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
    if (lane == 0) gsh[wave] = g;
    __syncthreads();
    const double gmean = (((gsh[0] + gsh[1]) + gsh[2]) + gsh[3]) / (double)n;
    for (int j = threadIdx.x; j < d; j += 256) {
        double s = 0.;
        for (int b = 0; b < nb; ++b) s += P[((size_t)b * d + i) * d + j];
        A[(size_t)i * d + j] = s / (double)n - gmean * W[(size_t)i * d + j];
    }
}

// after the polar factor W1 of A: lim = max_i | |sum_j W1[i][j] W[i][j]| - 1 | (scikit-learn's convergence measure) -> meas[k]
// (an atomic maximum of the bit patterns of non-negative doubles: order-independent; bf_ica_assemble_kernel zeroed the slot),
// the polar iteration's residual -> meas[n_meas + k], W1 -> Wbuf[k] and -> W.  A row per wave (a row's dot product reads only that
// row of W, which the wave then overwrites).
__global__ __launch_bounds__(256) void bf_ica_post_kernel(int d, const double *__restrict__ W1, double *__restrict__ W,
                                                         const double *__restrict__ resid, int k, int n_meas, double *__restrict__ Wbuf,
                                                         double *__restrict__ meas) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= d) return;
    double *dst = Wbuf + (size_t)k * d * d;
    double dot = 0.;
    for (int j = lane; j < d; j += 64) {
        const double v = W1[(size_t)i * d + j];
        dot += v * W[(size_t)i * d + j];
        dst[(size_t)i * d + j] = v;
    }
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    for (int j = lane; j < d; j += 64) W[(size_t)i * d + j] = W1[(size_t)i * d + j];
    if (lane == 0) {
        const double v = fabs(fabs(dot) - 1.);
        atomicMax((unsigned long long *)&meas[k], (unsigned long long)__double_as_longlong(v != v ? __builtin_inf() : v));
        if (i == 0) meas[n_meas + k] = resid[0];
    }
}

extern "C" int bfhip_ica_tanh(bfhip_ctx *ctx, long n, long n_pad, int d, double *y, double *partial) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 1 || n_pad < n || d < 1 || !y || !partial) return bf_set_error(BFHIP_ERR_ARG, "bfhip_ica_tanh: invalid argument");
    hipLaunchKernelGGL(bf_ica_tanh_kernel, dim3((unsigned)((n_pad + ICA_RB - 1) / ICA_RB)), dim3(256), 0, ctx->stream, n, n_pad, d, y, partial);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int bfhip_ica_assemble(bfhip_ctx *ctx, int d, int nb, const double *p, long n, long n_pad, const double *partial,
                                  const double *w, double *a, double *meas_k) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || d < 1 || nb < 1 || n < 1 || n_pad < n || !p || !partial || !w || !a)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_ica_assemble: invalid argument");
    hipLaunchKernelGGL(bf_ica_assemble_kernel, dim3(d), dim3(256), 0, ctx->stream, d, nb, p, n, (int)((n_pad + ICA_RB - 1) / ICA_RB), partial, w, a, meas_k);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int bfhip_ica_post(bfhip_ctx *ctx, int d, const double *w1, double *w, const double *resid, int k, int n_meas, double *wbuf,
                              double *meas) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || d < 1 || k < 0 || k >= n_meas || !w1 || !w || !resid || !wbuf || !meas)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_ica_post: invalid argument");
    hipLaunchKernelGGL(bf_ica_post_kernel, dim3((d + 3) / 4), dim3(256), 0, ctx->stream, d, w1, w, resid, k, n_meas, wbuf, meas);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void bf_zero_words_kernel(unsigned int *p, int n) {
    for (int e = threadIdx.x; e < n; e += blockDim.x) p[e] = 0u;
}

extern "C" int bfhip_polar_ns(bfhip_ctx *ctx, int d, const double *a, double *x, int n_iter, double *work, double *resid) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || d < 1 || d > 1024 || !a || !x || !work || !resid || n_iter < 0)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_polar_ns: invalid argument");
    const int nt = (d + 15) / 16;
    if (d <= 256 && bf_tune().polar_tiles != 1) {   // one launch, a workgroup per 16-row block, one grid barrier per step
        // the arrival counter and one residual slot per step live at the end of the CALLER's work array (not in the context's scratch,
        // which another call may reallocate: a HIP graph that holds this launch must stay valid)
        const size_t ws = 64 + (size_t)(n_iter + 2) * sizeof(unsigned long long);
        unsigned int *counter = (unsigned int *)(work + 2 * (size_t)d * d);
        hipLaunchKernelGGL(bf_zero_words_kernel, dim3(1), dim3(256), 0, ctx->stream, counter, (int)(ws / 4));
        const size_t lds = (size_t)16 * (16 * nt + 4) * sizeof(double);
        unsigned long long *slots = (unsigned long long *)((char *)counter + 64);
        if (nt <= BF_POLAR_LDS_MAXT && bf_tune().polar_tiles != 2) {
            const size_t lds_x = ((size_t)(16 * nt + 16) * (16 * nt + 4) + 2 * BF_POLAR_LDS_MAXT) * sizeof(double);   // 152 KB at d = 128
            static size_t lds_set = 0;
            if (lds_x > lds_set) {
                BF_HIP_CHECK(hipFuncSetAttribute((const void *)bf_polar_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x));
                lds_set = lds_x;
            }
            hipLaunchKernelGGL(bf_polar_lds_kernel, dim3(nt), dim3(64 * nt), lds_x, ctx->stream, d, a, x, n_iter, work, resid, counter, slots,
                               bf_tune().gstamps);
        }
        else if (nt <= 8)
            hipLaunchKernelGGL((bf_polar_rows_kernel<8, 32>), dim3(nt), dim3(64 * nt), lds, ctx->stream, d, a, x, n_iter, work, resid, counter, slots, bf_tune().gstamps);
        else
            hipLaunchKernelGGL((bf_polar_rows_kernel<16, 8>), dim3(nt), dim3(64 * nt), lds, ctx->stream, d, a, x, n_iter, work, resid, counter, slots,
                               bf_tune().gstamps);
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (d <= 512) {   // one launch, grid barriers between the products (every workgroup resident: at most 1024 single-wave workgroups)
        const size_t ws = 64 + (size_t)(n_iter + 2) * sizeof(unsigned long long);
        unsigned int *counter = (unsigned int *)(work + 2 * (size_t)d * d);
        hipLaunchKernelGGL(bf_zero_words_kernel, dim3(1), dim3(256), 0, ctx->stream, counter, (int)(ws / 4));   // (a kernel: a memset node of a HIP graph costs far more)
        hipLaunchKernelGGL(bf_polar_ns_kernel, dim3((nt * nt + 3) / 4), dim3(256), 0, ctx->stream, d, a, x, n_iter, work, resid, counter,
                           (unsigned long long *)((char *)counter + 64));
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    double *T = work, *Y = work + (size_t)d * d;   // X and Y take turns
    double *cur = (n_iter % 2 == 0) ? x : Y;        // (so that the last step writes x)
    hipLaunchKernelGGL(bf_ns_scale_kernel, dim3(1), dim3(256), 0, ctx->stream, d, a, cur);
    for (int it = 0; it < n_iter; ++it) {
        double *nxt = (cur == x) ? Y : x;
        hipLaunchKernelGGL(bf_small_gemm_kernel<true>, dim3(nt * nt), dim3(64), 0, ctx->stream, d, cur, cur, (const double *)NULL, T, 1., 0.);
        hipLaunchKernelGGL(bf_small_gemm_kernel<false>, dim3(nt * nt), dim3(64), 0, ctx->stream, d, T, cur, cur, nxt, -0.5, 1.5);
        cur = nxt;
    }
    hipLaunchKernelGGL(bf_small_gemm_kernel<true>, dim3(nt * nt), dim3(64), 0, ctx->stream, d, x, x, (const double *)NULL, T, 1., 0.);
    hipLaunchKernelGGL(bf_ns_resid_kernel, dim3(1), dim3(256), 0, ctx->stream, d, T, resid);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
