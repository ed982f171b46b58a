// bfhip_poly.hip -- multi-output PolyModel evaluation (modules/poly.py:430-503): linear, quadratic and cubic configs, and
// the chi-square stage that follows it in a pipeline (core/density.py:552-560).
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
    // cubic configs: per-output compact tables (same layouts as DevModel: A2, its transpose, and the symmetric fill of
    // the j < k < l coefficients stored [k][l][j]) and the dimension -> mask position maps
    const int n2 = (ds->cubic2 && ds->n2 > 0) ? ds->n2 : 0, n3 = (ds->cubic3 && ds->n3 > 0) ? ds->n3 : 0;
    if ((n2 && !ds->mask2) || (n3 && !ds->mask3)) return bf_set_error(BFHIP_ERR_ARG, "cubic configs need their masks");
    const size_t c2 = (size_t)m * n2 * n2, c3 = (size_t)m * n3 * n3 * n3;
    const size_t off_c = h.size();
    h.resize(off_c + 2 * c2 + c3, 0.);
    for (int o = 0; o < m; ++o) {
        for (int a = 0; a < n2; ++a)
            for (int b = 0; b < n2; ++b) {
                const double v = ds->cubic2[((size_t)o * n2 + a) * n2 + b];
                h[off_c + ((size_t)o * n2 + a) * n2 + b] = v;
                h[off_c + c2 + ((size_t)o * n2 + b) * n2 + a] = v;
            }
        double *T3 = h.data() + off_c + 2 * c2 + (size_t)o * n3 * n3 * n3;
        for (int a = 0; a < n3; ++a)
            for (int b = a + 1; b < n3; ++b)
                for (int c = b + 1; c < n3; ++c) {
                    const double v = ds->cubic3[(((size_t)o * n3 + a) * n3 + b) * n3 + c];
                    const int pp[3] = {a, b, c};
                    static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
                    for (int q = 0; q < 6; ++q) T3[((size_t)pp[perm[q][0]] * n3 + pp[perm[q][1]]) * n3 + pp[perm[q][2]]] = v;
                }
    }
    std::vector<int> hi((size_t)n2 + n3 + 2 * DP, -1);
    for (int a = 0; a < n2; ++a) { hi[a] = ds->mask2[a]; if (ds->mask2[a] < 0 || ds->mask2[a] >= d) return bf_set_error(BFHIP_ERR_ARG, "mask2 out of range"); hi[n2 + ds->mask2[a]] = a; }
    for (int a = 0; a < n3; ++a) { hi[n2 + DP + a] = ds->mask3[a]; if (ds->mask3[a] < 0 || ds->mask3[a] >= d) return bf_set_error(BFHIP_ERR_ARG, "mask3 out of range"); hi[n2 + DP + n3 + ds->mask3[a]] = a; }
    const size_t dbl_bytes = h.size() * sizeof(double);
    const size_t bytes = dbl_bytes + hi.size() * sizeof(int);
    if (ctx->pm_bytes < bytes) {
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->pm_buf) BF_HIP_CHECK(hipFree(ctx->pm_buf));
        ctx->pm_buf = NULL;
        ctx->pm_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->pm_buf, bytes));
        ctx->pm_bytes = bytes;
    }
    BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    BF_HIP_CHECK(hipMemcpy(ctx->pm_buf, h.data(), dbl_bytes, hipMemcpyHostToDevice));
    BF_HIP_CHECK(hipMemcpy((char *)ctx->pm_buf + dbl_bytes, hi.data(), hi.size() * sizeof(int), hipMemcpyHostToDevice));
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
    p.n2 = n2; p.n3 = n3;
    p.A2 = base + off_c; p.A2t = p.A2 + c2; p.T3t = p.A2t + c2;
    const int *ib = (const int *)((const char *)ctx->pm_buf + dbl_bytes);
    p.mask2 = ib; p.pos2 = ib + n2; p.mask3 = ib + n2 + DP; p.pos3 = ib + n2 + DP + n3;
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
    const bool has_cubic = pm.n2 > 0 || pm.n3 > 0;
    double *xst = lds + (STAGE ? MAT : 0) + wave * 16 * DP;  // this wave's points, plain layout (cubic configs)
    if (pm.use_bound) {  // the bound test of PolyModel._fun_and_jac: modules/poly.py:467-469
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
    if (has_cubic) {
#pragma unroll
        for (int e = 0; e < E; ++e) xst[c * DP + 4 * e + g] = xe[e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
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
        double quad = 0., lin = 0., dotj = 0., fcub = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double cl = pm.lin[(size_t)o * DP + 4 * e + g];
            quad += xe[e] * G[e];
            lin += cl * xe[e];
            G[e] += cl;                 // Jacobian row at the evaluation point
        }
        if (has_cubic) {  // cubic-2 / cubic-3 configs of this output (modules/_poly.pyx:49-137), compact tables
            DevModel cm;
            cm.DP = DP;
            cm.n2 = pm.n2; cm.n3 = pm.n3;
            cm.mask2 = pm.mask2; cm.pos2 = pm.pos2; cm.mask3 = pm.mask3; cm.pos3 = pm.pos3;
            cm.A2 = pm.A2 + (size_t)o * pm.n2 * pm.n2;
            cm.A2t = pm.A2t + (size_t)o * pm.n2 * pm.n2;
            cm.T3t = pm.T3t + (size_t)o * pm.n3 * pm.n3 * pm.n3;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double gc, fc;
                bf_cubic_grad(cm, 4 * e + g, xe[e], [&](int k) { return xst[c * DP + k]; }, gc, fc);
                G[e] += gc;
                fcub += fc;
            }
            fcub = bf_sum_g(fcub);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) dotj += G[e] * xm[e];
        quad = bf_sum_g(quad);
        lin = bf_sum_g(lin);
        double fo = ((pm.c0[o] + lin) + 0.5 * quad) + fcub;
        if (pm.use_bound) {
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
    const bool cubic = ctx->pm.n2 > 0 || ctx->pm.n3 > 0;
    const size_t lds = ((T <= 4 ? (size_t)256 * T * T : 0) + (cubic ? (size_t)4 * 16 * 16 * T : 0)) * sizeof(double);
    if (lds > 64 * 1024) BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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


// ---- chi-square stage of a surrogate pipeline (core/density.py:552-560 chain rule fused with the module) ------------
// One wave per point: lanes stride over the m outputs for r = prec (f - y) and the value, then over the d inputs for
// grad = -J^T r (coalesced reads of the Jacobian rows).
__global__ __launch_bounds__(256) void bf_chi2_stage_kernel(int n, int m, int d, const double *__restrict__ f, const double *__restrict__ jac,
                                                           const double *__restrict__ y, const double *__restrict__ prec,
                                                           const double *__restrict__ pdiag, double logp0, double *__restrict__ logp,
                                                           double *__restrict__ grad, double *__restrict__ rbuf) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pt = blockIdx.x * 4 + wv;
    if (pt >= n) return;
    const double *fp = f + (size_t)pt * m;
    double *r = rbuf + (size_t)pt * m;
    double acc = 0.;
    for (int o = lane; o < m; o += 64) {
        double ro;
        if (prec) {
            ro = 0.;
            for (int k = 0; k < m; ++k) ro += prec[(size_t)o * m + k] * (fp[k] - y[k]);
        } else {
            ro = pdiag[o] * (fp[o] - y[o]);
        }
        r[o] = ro;
        acc += (fp[o] - y[o]) * ro;
    }
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (lane == 0) logp[pt] = logp0 - 0.5 * acc;
    if (!grad) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double *jp = jac + (size_t)pt * m * d;
    for (int i = lane; i < d; i += 64) {
        double g = 0.;
        for (int o = 0; o < m; ++o) g += jp[(size_t)o * d + i] * r[o];
        grad[(size_t)pt * d + i] = -g;
    }
}

extern "C" int bfhip_chi2_stage(bfhip_ctx *ctx, int n, int m, int d, const double *f, const double *jac, const double *y,
                                const double *prec, const double *prec_diag, double logp0, double *logp, double *grad) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || m < 1 || d < 1 || !y || (!prec && !prec_diag) || (n > 0 && (!f || !logp)) || (grad && !jac))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_chi2_stage: invalid argument");
    if (n == 0) return 0;
    const size_t need = (size_t)n * m * sizeof(double);
    if (ctx->scratch_bytes < need) {
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) BF_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = NULL;
        ctx->scratch_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    hipLaunchKernelGGL(bf_chi2_stage_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, n, m, d, f, jac, y, prec, prec_diag, logp0,
                       logp, grad, (double *)ctx->scratch);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
