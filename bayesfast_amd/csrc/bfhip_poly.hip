// bfhip_poly.hip -- multi-output PolyModel evaluation (modules/poly.py:430-503) for linear + quadratic configs.
//
// Same wave-local MFMA layout as bfhip_eval.h (one wave = 16 points, lane (c, g) owns dimensions 4e + g); the
// outputs are a loop around the matvec: G_o = S_o x gives value and Jacobian row of output o.  A workgroup of four
// waves (64 points) stages S_o in LDS once per output (d <= 64).  The Jacobian (n, m, d) is the dominant traffic:
// 8 m d bytes per point written once.
#include <vector>
#include <cstring>
#include "bfhip_eval.h"

typedef bfhip_ctx::PolyDev PolyDev;

static int pm_padded_tiles(int d) { return d <= 16 ? 1 : d <= 32 ? 2 : d <= 64 ? 4 : 8; }

static void pm_to_fragments(const double *M, int d, int DP, double *frag) {
    const int T = DP / 16, NS = DP / 4;
    for (int t = 0; t < T; ++t)
        for (int s = 0; s < NS; ++s)
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * t + (l & 15), col = 4 * s + (l >> 4);
                frag[((size_t)t * NS + s) * 64 + l] = (row < d && col < d) ? M[(size_t)row * d + col] : 0.;
            }
}

extern "C" int bfhip_polymodel_upload(bfhip_ctx *ctx, const bfhip_polymodel_desc *ds) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || !ds) return bf_set_error(BFHIP_ERR_ARG, "bfhip_polymodel_upload: NULL argument");
    const int d = ds->d, m = ds->m;
    if (d < 1 || d > BFHIP_MAX_DIM || m < 1) return bf_set_error(BFHIP_ERR_ARG, "bfhip_polymodel_upload: d = %d, m = %d", d, m);
    if (!ds->c0 || !ds->lin) return bf_set_error(BFHIP_ERR_ARG, "bfhip_polymodel_upload: c0 and lin are required");
    if (ds->use_bound && (!ds->mu || !ds->hess || !ds->f_mu || !(ds->alpha > 0.)))
        return bf_set_error(BFHIP_ERR_ARG, "use_bound needs mu, hess, f_mu and alpha > 0");
    const int T = pm_padded_tiles(d), DP = 16 * T;
    const size_t MAT = (size_t)DP * DP;
    const size_t n_dbl = (ds->quad ? (size_t)m * MAT : 0) + (size_t)m * DP + 2 * (size_t)m + DP + MAT;
    std::vector<double> h(n_dbl, 0.);
    double *Sf = h.data(), *lin = Sf + (ds->quad ? (size_t)m * MAT : 0), *c0 = lin + (size_t)m * DP, *fmu = c0 + m,
           *mu = fmu + m, *Hf = mu + DP;
    std::vector<double> S((size_t)d * d);
    for (int o = 0; o < m; ++o) {
        c0[o] = ds->c0[o];
        if (ds->use_bound) fmu[o] = ds->f_mu[o];
        for (int i = 0; i < d; ++i) lin[(size_t)o * DP + i] = ds->lin[(size_t)o * d + i];
        if (ds->quad) {  // S = A + A^T from the upper triangle the reference reads (modules/_poly.pyx:13-43)
            const double *A = ds->quad + (size_t)o * d * d;
            for (int j = 0; j < d; ++j)
                for (int k = j; k < d; ++k) {
                    const double a = A[(size_t)j * d + k];
                    if (j == k) S[(size_t)j * d + j] = 2. * a;
                    else { S[(size_t)j * d + k] = a; S[(size_t)k * d + j] = a; }
                }
            pm_to_fragments(S.data(), d, DP, Sf + (size_t)o * MAT);
        }
    }
    if (ds->use_bound) {
        for (int i = 0; i < d; ++i) mu[i] = ds->mu[i];
        pm_to_fragments(ds->hess, d, DP, Hf);
    }
    const size_t bytes = n_dbl * sizeof(double);
    if (ctx->pm_bytes < bytes) {
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->pm_buf) BF_HIP_CHECK(hipFree(ctx->pm_buf));
        ctx->pm_buf = NULL;
        ctx->pm_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->pm_buf, bytes));
        ctx->pm_bytes = bytes;
    }
    BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    BF_HIP_CHECK(hipMemcpy(ctx->pm_buf, h.data(), bytes, hipMemcpyHostToDevice));
    PolyDev &p = ctx->pm;
    memset(&p, 0, sizeof(p));
    p.d = d; p.DP = DP; p.m = m; p.use_bound = ds->use_bound != 0; p.has_quad = ds->quad != NULL;
    const double *base = (const double *)ctx->pm_buf;
    p.Sf = base;
    p.lin = base + (ds->quad ? (size_t)m * MAT : 0);
    p.c0 = p.lin + (size_t)m * DP;
    p.f_mu = p.c0 + m;
    p.mu = p.f_mu + m;
    p.Hf = p.mu + DP;
    p.alpha = ds->alpha;
    ctx->has_pm = 1;
    return 0;
}

template <int T>
__global__ __launch_bounds__(256) void bf_polymodel_eval_kernel(PolyDev pm, int n, const double *__restrict__ x,
                                                               double *__restrict__ f, double *__restrict__ jac) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int E = 4 * T, DP = 16 * T, MAT = DP * DP;
    constexpr bool STAGE = T <= 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x * 4 + wave;
    const int pt = tile * 16 + c;
    const int d = pm.d, m = pm.m;
    const bool live = pt < n;
    double xv[E], xm[E], hv[E], xe[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int dim = 4 * e + g;
        xv[e] = (live && dim < d) ? x[(size_t)pt * d + dim] : 0.;
        xe[e] = xv[e];
        xm[e] = 0.;
        hv[e] = 0.;
    }
    double beta = 0.;
    bool oob = false;
    if (pm.use_bound && pm.has_quad) {  // the bound test of PolyModel._fun_and_jac: modules/poly.py:467-469
        double b2 = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) xm[e] = xv[e] - pm.mu[4 * e + g];
        bf_matvec_w1<T>(pm.Hf, xm, hv, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) b2 += xm[e] * hv[e];
        b2 = bf_sum_g(b2);
        beta = sqrt(b2);
        oob = beta > pm.alpha;
        if (oob) {
#pragma unroll
            for (int e = 0; e < E; ++e) xe[e] = (pm.alpha * xv[e] + (beta - pm.alpha) * pm.mu[4 * e + g]) / beta;  // :482
        }
    }
    // the outputs are split over blockIdx.y so that small batches still fill the chip (the bound test above is
    // repeated per chunk; it is one matvec)
    const int o_per = (m + gridDim.y - 1) / gridDim.y;
    const int o_beg = blockIdx.y * o_per, o_end = min(m, o_beg + o_per);
    for (int o = o_beg; o < o_end; ++o) {
        const double *Sf = pm.Sf + (size_t)o * MAT;
        if (STAGE && pm.has_quad) {
            __syncthreads();
            for (int i = threadIdx.x; i < MAT / 2; i += 256) ((d2_t *)lds)[i] = ((const d2_t *)Sf)[i];
            __syncthreads();
            Sf = lds;
        }
        double G[E];
        if (pm.has_quad) {
            bf_matvec_w1<T>(Sf, xe, G, lane);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) G[e] = 0.;
        }
        double quad = 0., lin = 0., dotj = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double cl = pm.lin[(size_t)o * DP + 4 * e + g];
            quad += xe[e] * G[e];
            lin += cl * xe[e];
            G[e] += cl;                 // Jacobian row at the evaluation point
            dotj += G[e] * xm[e];
        }
        quad = bf_sum_g(quad);
        lin = bf_sum_g(lin);
        double fo = (pm.c0[o] + lin) + 0.5 * quad;
        if (pm.use_bound && pm.has_quad) {
            dotj = bf_sum_g(dotj);
            if (oob) {  // linear extrapolation from the projected point: modules/poly.py:484-503
                const double f0 = fo;
                fo = (beta * f0 - (beta - pm.alpha) * pm.f_mu[o]) / pm.alpha;
                const double coef = (f0 - pm.f_mu[o]) / pm.alpha - dotj / beta;
#pragma unroll
                for (int e = 0; e < E; ++e) G[e] = G[e] + coef * (hv[e] / beta);
            }
        }
        if (live) {
            if (g == 0) f[(size_t)pt * m + o] = fo;
            if (jac) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int dim = 4 * e + g;
                    if (dim < d) jac[((size_t)pt * m + o) * d + dim] = G[e];
                }
            }
        }
    }
}

template <int T>
static int launch_polymodel_eval(bfhip_ctx *ctx, int n, const double *x, double *f, double *jac) {
    auto k = bf_polymodel_eval_kernel<T>;
    const size_t lds = T <= 4 ? (size_t)256 * T * T * sizeof(double) : 0;
    const int grid = (n + 63) / 64;
    int ny = (4 * ctx->n_cu + grid - 1) / grid;  // aim at about four workgroups per CU
    if (ny > ctx->pm.m) ny = ctx->pm.m;
    if (ny < 1) ny = 1;
    hipLaunchKernelGGL(k, dim3(grid, ny), dim3(256), lds, ctx->stream, ctx->pm, n, x, f, jac);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int bfhip_polymodel_eval(bfhip_ctx *ctx, int n, const double *x, double *f, double *jac) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || (n > 0 && (!x || !f))) return bf_set_error(BFHIP_ERR_ARG, "bfhip_polymodel_eval: invalid argument");
    if (!ctx->has_pm) return bf_set_error(BFHIP_ERR_STATE, "bfhip_polymodel_eval: no polymodel uploaded");
    if (n == 0) return 0;
    switch (ctx->pm.DP / 16) {
    case 1: return launch_polymodel_eval<1>(ctx, n, x, f, jac);
    case 2: return launch_polymodel_eval<2>(ctx, n, x, f, jac);
    case 4: return launch_polymodel_eval<4>(ctx, n, x, f, jac);
    case 8: return launch_polymodel_eval<8>(ctx, n, x, f, jac);
    }
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "unsupported padded dimension %d", ctx->pm.DP);
}
