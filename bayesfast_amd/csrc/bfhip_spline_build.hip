// bfhip_spline_build.hip -- the Gaussianizing splines of one SIT iteration built ON THE DEVICE, one workgroup per coordinate
// (reference: SIT._gaussianize_1d, transforms/sit.py:223-227, which calls cubic_spline,
// utils/cubic.py:19-260, with fun = norm.ppf(kde.cdf(.)), utils/kde.py:322-354).
//
// The construction is control logic on ~100-300 knots around a handful of function evaluations (the KDE cdf: a sum over all n
// samples per point).  Rounds 3-5 ran the logic on the host with the evaluations batched on the device: eight round trips per
// iteration and 0.6 ms of NumPy per spline (0.5 s of a config-5 GBS run).  Here a workgroup of 1024 threads owns a coordinate:
// all threads take the cdf sums, thread 0 (or a thread per knot) runs the logic, nothing leaves the chip until the finished knots,
// values and coefficient rows are written.  The arithmetic follows NumPy's / LAPACK's statement by statement where the result
// decides something (percentile interpolation, linspace, the pairwise sums of np.sum / np.mean, dgtsv's elimination order), so
// that knot sets are the host construction's (bayesfast_amd/utils/spline.py, itself pinned to the reference's fixtures); the cdf
// sums differ from bfhip_kde_cdf's in summation order only (1e-16 relative).  This file is compiled with -ffp-contract=off.
#include <cmath>
#include "bfhip_common.h"
#include "bfhip_ndtri.h"

#define SB_TH 1024
#define SB_MAXK 512     // knots a spline may reach (99 to start with; a spline that needs more is reported, the host builds it)
#define SB_KP 8         // points a thread keeps in registers per pass over the samples

struct SbOpts {
    int n_grid, edge_bins, n_inner, split, max_add, stride;
    double max_width;
};

struct SbShared {
    double x[SB_MAXK], y[SB_MAXK], c[(SB_MAXK + 1) * 4];
    double pts[SB_MAXK], vals[SB_MAXK];
    double dl[SB_MAXK], dd[SB_MAXK], du[SB_MAXK], rhs[SB_MAXK];
    double red[(SB_TH / 64) * SB_KP];
    unsigned char good[SB_MAXK];
    int n, m, flag, all_good;
    double k_left, k_right;
};

// np.add.reduce of a contiguous array (np.sum, the numerator of np.mean): NumPy's pairwise summation
__device__ double sb_np_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return sb_np_sum(a, n2) + sb_np_sum(a + n2, n - n2);
}

// np.percentile(xs[0:n] - shift, q) for sorted xs (NumPy's default linear method: utils/spline.py percentile_sorted)
__device__ inline double sb_percentile(const double *xs, long n, double q, double shift) {
    const double quant = q / 100.;
    const double virt = (double)(n - 1) * quant;
    if (virt >= (double)(n - 1)) return xs[n - 1] - shift;
    if (virt < 0.) return xs[0] - shift;
    const double fl = floor(virt);
    const long lo = (long)fl;
    const double t = virt - fl;
    const double a = xs[lo] - shift, b = xs[lo + 1] - shift;
    const double step = b - a;
    return (t >= 0.5) ? b - step * (1. - t) : a + step * t;
}

// vals[i] = ndtri(sum_k w[k] ndtr((pts[i] - data[k]) / h)) for i < m: every thread strides over the samples with SB_KP points in
// registers; wave sums by butterflies, the sixteen wave sums added in order by one thread per point
__device__ void sb_eval(SbShared &S, int m, const double *__restrict__ data, const double *__restrict__ w, long n, double inv) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i0 = 0; i0 < m; i0 += SB_KP) {
        double p[SB_KP], acc[SB_KP];
#pragma unroll
        for (int t = 0; t < SB_KP; ++t) {
            p[t] = S.pts[(i0 + t < m) ? i0 + t : m - 1];
            acc[t] = 0.;
        }
        for (long k = tid; k < n; k += SB_TH) {
            const double xk = data[k], wk = 0.5 * w[k];
#pragma unroll
            for (int t = 0; t < SB_KP; ++t) acc[t] += wk * erfc((xk - p[t]) * inv);
        }
#pragma unroll
        for (int t = 0; t < SB_KP; ++t) {
            double v = acc[t];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) S.red[wave * SB_KP + t] = v;
        }
        __syncthreads();
        if (tid < SB_KP && i0 + tid < m) {
            double s = 0.;
            for (int q = 0; q < SB_TH / 64; ++q) s += S.red[q * SB_KP + tid];
            S.vals[i0 + tid] = bf_ndtri(s);
        }
        __syncthreads();
    }
}

// np.insert(x, np.searchsorted(x, new), new) and the same for y: new (sorted) merged into the knots, a new point before the
// knots that are >= it.  One thread; S.pts / S.vals hold the m new points and their values.
__device__ void sb_merge(SbShared &S, int m) {
    int i = S.n - 1, j = m - 1, o = S.n + m - 1;
    while (j >= 0) {
        if (i >= 0 && S.x[i] >= S.pts[j]) { S.x[o] = S.x[i]; S.y[o] = S.y[i]; --i; }
        else { S.x[o] = S.pts[j]; S.y[o] = S.vals[j]; --j; }
        --o;
    }
    S.n += m;
}

// the clamped C2 spline (utils/cubic.py:142-184): tridiagonal system for the knot slopes, solved as LAPACK's dgtsv does (partial
// pivoting between neighbouring rows, its elimination and back-substitution order), then the coefficient rows
__device__ void sb_fit(SbShared &S) {
    const int n = S.n, tid = threadIdx.x;
    // rows: dl = sub-diagonal (n-1), dd = diagonal (n), du = super-diagonal (n-1)
    for (int i = tid; i < n; i += SB_TH) {
        if (i == 0 || i == n - 1) {
            S.dd[i] = 1.;
            S.rhs[i] = (i == 0) ? S.k_left : S.k_right;
            if (i == 0) S.du[0] = 0.;
            if (i == n - 1) S.dl[n - 2] = 0.;
        } else {
            const double w0 = S.x[i] - S.x[i - 1], w1 = S.x[i + 1] - S.x[i];
            const double ch0 = (S.y[i] - S.y[i - 1]) / w0, ch1 = (S.y[i + 1] - S.y[i]) / w1;
            S.dd[i] = 2 * (w0 + w1);
            S.du[i] = w0;          // band[0, 2:] = w[:-1]: above the diagonal of row i sits w[i - 1]
            S.dl[i - 1] = w1;      // band[2, :-2] = w[1:]: below the diagonal of column i - 1 (row i) sits w[i]
            S.rhs[i] = 3 * (w1 * ch0 + w0 * ch1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        double *dl = S.dl, *d = S.dd, *du = S.du, *b = S.rhs;
        bool singular = false;
        for (int i = 0; i < n - 2 && !singular; ++i) {
            if (fabs(d[i]) >= fabs(dl[i])) {
                if (d[i] != 0.) {
                    const double fact = dl[i] / d[i];
                    d[i + 1] = d[i + 1] - fact * du[i];
                    b[i + 1] = b[i + 1] - fact * b[i];
                } else singular = true;
                dl[i] = 0.;
            } else {
                const double fact = d[i] / dl[i];
                d[i] = dl[i];
                double temp = d[i + 1];
                d[i + 1] = du[i] - fact * temp;
                dl[i] = du[i + 1];
                du[i + 1] = -fact * dl[i];
                du[i] = temp;
                temp = b[i];
                b[i] = b[i + 1];
                b[i + 1] = temp - fact * b[i + 1];
            }
        }
        if (n > 1 && !singular) {
            const int i = n - 2;
            if (fabs(d[i]) >= fabs(dl[i])) {
                if (d[i] != 0.) {
                    const double fact = dl[i] / d[i];
                    d[i + 1] = d[i + 1] - fact * du[i];
                    b[i + 1] = b[i + 1] - fact * b[i];
                } else singular = true;
            } else {
                const double fact = d[i] / dl[i];
                d[i] = dl[i];
                double temp = d[i + 1];
                d[i + 1] = du[i] - fact * temp;
                du[i] = temp;
                temp = b[i];
                b[i] = b[i + 1];
                b[i + 1] = temp - fact * b[i + 1];
            }
        }
        if (singular || d[n - 1] == 0.) S.flag |= 2;
        else {
            b[n - 1] = b[n - 1] / d[n - 1];
            if (n > 1) b[n - 2] = (b[n - 2] - du[n - 2] * b[n - 1]) / d[n - 2];
            for (int i = n - 3; i >= 0; --i) b[i] = (b[i] - du[i] * b[i + 1] - dl[i] * b[i + 2]) / d[i];
        }
    }
    __syncthreads();
    const double *s = S.rhs;
    for (int i = tid; i <= n; i += SB_TH) {
        double *c = S.c + 4 * i;
        if (i == 0) { c[0] = 0.; c[1] = 0.; c[2] = S.k_left; c[3] = S.y[0]; }
        else if (i == n) { c[0] = 0.; c[1] = 0.; c[2] = S.k_right; c[3] = S.y[n - 1]; }
        else {
            const double w = S.x[i] - S.x[i - 1], chord = (S.y[i] - S.y[i - 1]) / w;
            const double t = (s[i - 1] + s[i] - 2 * chord) / w;
            c[0] = t / w;
            c[1] = (chord - s[i - 1]) / w - t;
            c[2] = s[i - 1];
            c[3] = S.y[i - 1];
        }
    }
    __syncthreads();
}

// one flag per interior interval i = 0 .. n-2 (coefficient row i + 1): monotone increasing? (utils/_cubic.pyx:166-186, 336-343)
__device__ void sb_flags(SbShared &S) {
    const int n = S.n, tid = threadIdx.x;
    if (tid == 0) S.all_good = 1;
    __syncthreads();
    for (int i = tid; i < n - 1; i += SB_TH) {
        const double *c = S.c + 4 * (i + 1);
        const double w = S.x[i + 1] - S.x[i], c0 = c[0], c1 = c[1], c2 = c[2];
        const double slope_r = 3 * c0 * w * w + 2 * c1 * w + c2, bend_r = 3 * c0 * w + c1, disc = c1 * c1 - 3 * c0 * c2;
        bool ok = (c2 > 0) && (slope_r > 0) && (c1 * bend_r >= 0);
        ok = ok || ((c0 > 0) && (disc < 0));
        S.good[i] = ok ? 1 : 0;
        if (!ok) S.all_good = 0;   // (a benign race: every writer stores 0)
    }
    __syncthreads();
}

// utils/cubic.py:190-216: runs of (almost) non-increasing values become the straight line across them.  One thread.
__device__ void sb_straighten(SbShared &S) {
    const int n = S.n, nk = n - 1;
    double thr = 1e-10;
    for (;;) {
        // bad = flatnonzero(diff(y) / diff(x) < thr), taken before this pass changes anything
        int nb = 0;
        int *bad = (int *)S.dl;   // (scratch: the solver's rows are rebuilt by the next fit)
        for (int i = 0; i < nk; ++i)
            if ((S.y[i + 1] - S.y[i]) / (S.x[i + 1] - S.x[i]) < thr) bad[nb++] = i;
        if (nb == 0) return;
        int b0 = 0;
        while (b0 < nb) {
            int i = b0;
            int start = bad[b0] - 1;              // (np.max(bad[0] - 1, 0) of the reference is bad[0] - 1; -1 indexes from the end)
            while (i < nb - 1 && bad[i + 1] - bad[i] <= 2) ++i;
            int end = bad[i] + 1;
            if (end > nk - 1) end = nk - 1;
            const int si = (start < 0) ? start + n : start;
            const double line = (S.y[end + 1] - S.y[si]) / (S.x[end + 1] - S.x[si]);
            for (int j = start + 1; j <= end; ++j) S.y[j] = S.y[si] + line * (S.x[j] - S.x[si]);
            b0 = i + 1;
        }
        thr = 1e-8;
    }
}

__global__ __launch_bounds__(SB_TH) void bf_spline_build_kernel(long n, const double *__restrict__ sorted, const double *__restrict__ data,
                                                               const double *__restrict__ w, const double *__restrict__ h,
                                                               const double *__restrict__ grid, const double *__restrict__ inner, SbOpts o,
                                                               double *__restrict__ out_x, double *__restrict__ out_y,
                                                               double *__restrict__ out_c, int *__restrict__ out_n) {
    __shared__ SbShared S;
    const int j = blockIdx.x, tid = threadIdx.x;
    const double *xs = sorted + (size_t)j * n, *dj = data + (size_t)j * n;
    const double inv = 0.70710678118654752440 / h[j];
    if (tid == 0) { S.flag = 0; S.n = 0; }
    // 1. knots: distinct percentiles on the grid
    for (int i = tid; i < o.n_grid; i += SB_TH) S.pts[i] = sb_percentile(xs, n, grid[i], 0.);
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int i = 0; i < o.n_grid; ++i)
            if (m == 0 || S.pts[i] != S.x[m - 1]) S.x[m++] = S.pts[i];
        S.n = m;
        if (m < 2 * o.edge_bins + 3) S.flag |= 1;
    }
    __syncthreads();
    if (S.flag) { if (tid == 0) out_n[2 * j] = 0, out_n[2 * j + 1] = S.flag; return; }
    for (int i = tid; i < S.n; i += SB_TH) S.pts[i] = S.x[i];
    __syncthreads();
    sb_eval(S, S.n, dj, w, n, inv);
    for (int i = tid; i < S.n; i += SB_TH) S.y[i] = S.vals[i];
    __syncthreads();
    // 2. the two end slopes: least squares through the outermost knot on percentile points of the samples beyond knot edge_bins
    //    from either end
    __shared__ long s_below, s_above;
    __shared__ double s_t[2 * 128];
    const int ne = o.n_inner;
    if (tid == 0) {
        const double kl = S.x[o.edge_bins], kr = S.x[S.n - o.edge_bins - 1];
        long lo = 0, hi = n;                      // searchsorted(xs, kl, 'left')
        while (lo < hi) { const long mid = (lo + hi) >> 1; if (xs[mid] < kl) lo = mid + 1; else hi = mid; }
        s_below = lo;
        lo = 0, hi = n;                           // searchsorted(xs, kr, 'right')
        while (lo < hi) { const long mid = (lo + hi) >> 1; if (xs[mid] <= kr) lo = mid + 1; else hi = mid; }
        s_above = lo;
        if (s_below < 1 || s_above > n - 1) S.flag |= 1;
    }
    __syncthreads();
    if (S.flag) { if (tid == 0) out_n[2 * j] = 0, out_n[2 * j + 1] = S.flag; return; }
    for (int i = tid; i < 2 * ne; i += SB_TH) {
        const bool left = i < ne;
        const double knot = left ? S.x[0] : S.x[S.n - 1];
        const double t = left ? sb_percentile(xs, s_below, inner[i], knot) : sb_percentile(xs + s_above, n - s_above, inner[i - ne], knot);
        s_t[i] = t;
        S.pts[i] = t + knot;
    }
    __syncthreads();
    sb_eval(S, 2 * ne, dj, w, n, inv);
    if (tid < 2) {
        const bool left = tid == 0;
        const double value = left ? S.y[0] : S.y[S.n - 1];
        double *tf = left ? S.dl : S.dd, *tt = left ? S.du : S.rhs;
        const double *t = s_t + (left ? 0 : ne), *v = S.vals + (left ? 0 : ne);
        for (int i = 0; i < ne; ++i) { tf[i] = t[i] * (v[i] - value); tt[i] = t[i] * t[i]; }
        const double k = sb_np_sum(tf, ne) / sb_np_sum(tt, ne);
        if (left) S.k_left = k; else S.k_right = k;
    }
    __syncthreads();
    // 3. interior gaps wider than max_width mean gaps get equally spaced extra knots
    if (tid == 0) {
        const int nk = S.n - 1;
        double *rel = S.dl;
        for (int i = 0; i < nk; ++i) rel[i] = S.x[i + 1] - S.x[i];
        const double mean = sb_np_sum(rel, nk) / (double)nk;
        for (int i = 0; i < nk; ++i) rel[i] = rel[i] / mean;
        int first = 0;
        while (rel[first] > o.max_width) { ++first; if (first >= S.n - 2) break; }
        int last = S.n - 2;
        while (rel[last] > o.max_width) { --last; if (last <= 0) break; }
        int m = 0;
        if (first > last) S.flag |= 4;
        else
            for (int g = first; g <= last; ++g)
                if (rel[g] > o.max_width) {
                    const int div = (int)ceil(rel[g] / (double)o.split);   // np.linspace(a, b, div + 1)[1:-1]
                    const double a = S.x[g], b = S.x[g + 1], step = (b - a) / (double)div;
                    for (int q = 1; q < div; ++q) {
                        if (m >= SB_MAXK) { S.flag |= 8; break; }
                        S.pts[m++] = (step == 0.) ? ((double)q / (double)div) * (b - a) + a : (double)q * step + a;
                    }
                }
        if (S.n + m > SB_MAXK) S.flag |= 8;
        S.m = m;
    }
    __syncthreads();
    if (S.flag) { if (tid == 0) out_n[2 * j] = 0, out_n[2 * j + 1] = S.flag; return; }
    if (S.m) {
        sb_eval(S, S.m, dj, w, n, inv);
        if (tid == 0) sb_merge(S, S.m);
        __syncthreads();
    }
    // 4./5. fit; while some interval is not monotone, split the offending intervals and refit
    sb_fit(S);
    sb_flags(S);
    for (int rounds = 0; !S.all_good && rounds < o.max_add && !S.flag; ++rounds) {
        if (tid == 0) {
            int m = 0;
            for (int i = 0; i < S.n - 1; ++i)
                if (!S.good[i]) {
                    const double lo = S.x[i], hi = S.x[i + 1], st = (hi - lo) / (double)o.split;
                    for (int q = 1; q < o.split; ++q) {
                        if (m >= SB_MAXK) { S.flag |= 8; break; }
                        S.pts[m++] = (double)q * st + lo;
                    }
                }
            if (S.n + m > SB_MAXK) S.flag |= 8;
            S.m = m;
        }
        __syncthreads();
        if (S.flag) break;
        sb_eval(S, S.m, dj, w, n, inv);
        if (tid == 0) {
            sb_merge(S, S.m);
            if (rounds == o.max_add - 1) sb_straighten(S);
        }
        __syncthreads();
        sb_fit(S);
        sb_flags(S);
    }
    if (S.flag) { if (tid == 0) out_n[2 * j] = 0, out_n[2 * j + 1] = S.flag; return; }
    if (!S.all_good) {   // utils/cubic.py:128-136: straight segments where the cubic still turns
        for (int i = tid; i < S.n - 1; i += SB_TH)
            if (!S.good[i]) {
                double *c = S.c + 4 * (i + 1);
                c[0] = 0.; c[1] = 0.; c[2] = (S.y[i + 1] - S.y[i]) / (S.x[i + 1] - S.x[i]); c[3] = S.y[i];
            }
        __syncthreads();
        sb_flags(S);
        if (tid == 0 && !S.all_good) S.flag |= 16;   // 'Not all the intervals are monotone.' (a warning, not a failure)
        __syncthreads();
    }
    for (int i = tid; i < S.n; i += SB_TH) {
        out_x[(size_t)j * o.stride + i] = S.x[i];
        out_y[(size_t)j * o.stride + i] = S.y[i];
    }
    for (int i = tid; i < 4 * (S.n + 1); i += SB_TH) out_c[(size_t)j * 4 * (o.stride + 1) + i] = S.c[i];
    if (tid == 0) { out_n[2 * j] = S.n; out_n[2 * j + 1] = S.flag; }
}

extern "C" int bfhip_spline_build(bfhip_ctx *ctx, int d, long n, const double *sorted, const double *data, const double *w, const double *h,
                                  int n_grid, const double *grid, int edge_bins, int n_inner, const double *inner, double max_width,
                                  int split, int max_add, int stride, double *out_x, double *out_y, double *out_c, int *out_n) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || d < 1 || n < 2 || !sorted || !data || !w || !h || !grid || !inner || !out_x || !out_y || !out_c || !out_n)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_spline_build: invalid argument");
    if (n_grid < 3 || n_grid > SB_MAXK || edge_bins < 0 || n_inner < 1 || n_inner > 128 || split < 2 || max_add < 0 || stride < SB_MAXK ||
        !(max_width > 0.))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_spline_build: options out of range (grid <= 512 points, edge points <= 128, stride >= 512)");
    SbOpts o;
    o.n_grid = n_grid, o.edge_bins = edge_bins, o.n_inner = n_inner, o.split = split, o.max_add = max_add, o.stride = stride;
    o.max_width = max_width;
    hipLaunchKernelGGL(bf_spline_build_kernel, dim3(d), dim3(SB_TH), 0, ctx->stream, n, sorted, data, w, h, grid, inner, o, out_x, out_y, out_c,
                       out_n);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
