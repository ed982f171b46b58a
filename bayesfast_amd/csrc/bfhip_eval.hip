// bfhip_eval.hip -- batched surrogate logp+grad and batched leapfrog step (wave-local MFMA layout).
#include "bfhip_eval.h"
#include "bfhip_sampler_defs.h"

// Stages the model's fragments and per-dimension table into LDS (coefficient matrices are read by
// every MFMA of every wave; 32 KB each at DP = 64).  Returns pointers valid after __syncthreads().
struct StagedModel {
    const double *Sf, *Hf, *Hdf, *pd;
};

template <int T, bool STAGE>
__device__ inline StagedModel bf_stage_model(const DevModel &m, double *lds) {
    constexpr int DP = 16 * T;
    constexpr int MAT = DP * DP;
    StagedModel s;
    if constexpr (!STAGE) {
        s.Sf = m.Sf; s.Hf = m.Hf; s.Hdf = m.Hdf; s.pd = m.pd;
        return s;
    } else {
        double *p = lds;
        double *pdl = p; p += PD_N * DP;
        double *S = p; if (m.has_quad) p += MAT;
        double *H = p; if (m.use_bound) p += MAT;
        double *Hd = p;
        const int tid = threadIdx.x, nt = blockDim.x;
        for (int i = tid; i < PD_N * DP; i += nt) pdl[i] = m.pd[i];
        // 16-byte copies, coalesced
        const d2_t *src; d2_t *dst;
        if (m.has_quad) { src = (const d2_t *)m.Sf; dst = (d2_t *)S; for (int i = tid; i < MAT / 2; i += nt) dst[i] = src[i]; }
        if (m.use_bound) { src = (const d2_t *)m.Hf; dst = (d2_t *)H; for (int i = tid; i < MAT / 2; i += nt) dst[i] = src[i]; }
        if (m.use_decay) { src = (const d2_t *)m.Hdf; dst = (d2_t *)Hd; for (int i = tid; i < MAT / 2; i += nt) dst[i] = src[i]; }
        __syncthreads();
        s.Sf = S; s.Hf = H; s.Hdf = Hd; s.pd = pdl;
        return s;
    }
}

static size_t bf_stage_bytes(const DevModel &m) {  // bytes of the staged model (0 when it is not staged)
    if (m.DP > 64) return 0;
    size_t mat = (size_t)m.DP * m.DP * sizeof(double);
    return (size_t)PD_N * m.DP * sizeof(double) + mat * ((m.has_quad ? 1 : 0) + (m.use_bound ? 1 : 0) + (m.use_decay ? 1 : 0));
}
static size_t bf_eval_lds_bytes(const DevModel &m) {  // + the wave-private coordinate stage of the cubic terms
    return bf_stage_bytes(m) + (m.has_cubic ? (size_t)4 * 16 * m.DP * sizeof(double) : 0);
}

// Density.logp_and_grad over n points; replaces the per-row Python recursion of core/density.py:523-525.
template <int T, bool STAGE, bool PL>
__global__ __launch_bounds__(256) void bf_logp_grad_kernel(DevModel m, int stage_dbl, int n, const double *__restrict__ x,
                                                           int original_space, double *__restrict__ logp,
                                                           double *__restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int E = 4 * T;
    StagedModel sm = bf_stage_model<T, STAGE>(m, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    double *xst = lds + stage_dbl + wave * 16 * 16 * T;
    const int n_tiles = (n + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int pt = tile * 16 + c;
        double xv[E], gv[E], lp;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = 4 * e + g;
            xv[e] = (pt < n && dim < m.d) ? x[(size_t)pt * m.d + dim] : 0.;
        }
        bf_eval_w1<T, PL>(m, sm.Sf, sm.Hf, sm.Hdf, sm.pd, original_space, xv, lp, gv, lane, xst);
        if (pt < n) {
            if (g == 0) logp[pt] = lp;
            if (grad) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int dim = 4 * e + g;
                    if (dim < m.d) grad[(size_t)pt * m.d + dim] = gv[e];
                }
            }
        }
    }
}

// d = 128, the common surrogate (linear + quadratic configs with the bound, nothing else: bf_model_plain): one WORKGROUP of
// eight waves evaluates 16 points.  Wave w owns row tile w of S X^T and H (X - mu)^T and dimensions 16 w .. 16 w + 15 of
// the 16 points -- lane (c, g) element r is dimension 16 w + 4 r + g of point c, so the MFMA result lands on the lane
// that owns it, as in the wave-local layout -- keeps the 2 x 32 A fragments of its tile in registers for the whole
// launch, and the points' coordinates cross the waves through LDS as B operands.  (The wave-local kernel above holds all
// 128 dimensions of 16 points in ONE wave: 32 elements per lane for every vector of the evaluation and 256 operand loads
// per matvec, 117-243 spilled VGPRs, 7 % of the FP64 MFMA rate.)  The sums over dimensions are taken per wave and then
// over the eight waves in wave order: every wave holds the same totals and takes the same decisions.
struct Coop128 {
    static constexpr int NS = 32;
    double afS[NS], afH[NS];   // A fragments of this wave's row tile
    double c_lin[4], c_mu[4];
    int lane, w, c, g;
};
__device__ __forceinline__ void bf_coop128_init(Coop128 &k, const DevModel &m) {
    k.lane = threadIdx.x & 63;
    k.w = threadIdx.x >> 6;
    k.c = k.lane & 15;
    k.g = k.lane >> 4;
#pragma unroll
    for (int s = 0; s < Coop128::NS; ++s) {
        k.afS[s] = m.Sf[(size_t)(k.w * Coop128::NS + s) * 64 + k.lane];
        k.afH[s] = m.Hf[(size_t)(k.w * Coop128::NS + s) * 64 + k.lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int dim = 16 * k.w + 4 * r + k.g;
        k.c_lin[r] = m.pd[PD_LIN * 128 + dim];
        k.c_mu[r] = m.pd[PD_MU * 128 + dim];
    }
}
__device__ __forceinline__ d4_t bf_coop128_matvec(const double (&af)[Coop128::NS], const double (*xb)[64], int lane) {
    d4_t acc = {0., 0., 0., 0.};
#pragma unroll
    for (int s = 0; s < Coop128::NS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], xb[s][lane], acc, 0, 0, 0);
    return acc;
}
// the eight waves' partial sums of point c, in wave order
__device__ __forceinline__ double bf_coop128_total(const double (*rb)[16], int c) {
    double t = rb[0][c];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += rb[j][c];
    return t;
}
// logp and gradient (this lane's four dimensions) of the workgroup's 16 points at xv; contains workgroup barriers: the
// first one also separates this evaluation from whatever used XB / RB before it
__device__ __forceinline__ void bf_coop128_eval(const Coop128 &k, const DevModel &m, double (*XB)[Coop128::NS][64],
                                                double (*RB)[8][16], const double (&xv)[4], double &f, double (&gv)[4]) {
    const int lane = k.lane, w = k.w, c = k.c, g = k.g;
    double xm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xm[r] = xv[r] - k.c_mu[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        XB[0][4 * w + r][lane] = xv[r];
        XB[1][4 * w + r][lane] = xm[r];
    }
    __syncthreads();
    const d4_t gS = bf_coop128_matvec(k.afS, XB[0], lane), gH = bf_coop128_matvec(k.afH, XB[1], lane);
    double quad = 0., lin = 0., b2 = 0.;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        quad += xv[r] * gS[r];
        lin += k.c_lin[r] * xv[r];
        b2 += xm[r] * gH[r];
        gv[r] = gS[r] + k.c_lin[r];
    }
    quad = bf_sum_g(quad); lin = bf_sum_g(lin); b2 = bf_sum_g(b2);
    if (g == 0) { RB[0][w][c] = quad; RB[1][w][c] = lin; RB[2][w][c] = b2; }
    __syncthreads();
    quad = bf_coop128_total(RB[0], c); lin = bf_coop128_total(RB[1], c); b2 = bf_coop128_total(RB[2], c);
    f = (m.c0 + lin) + 0.5 * quad;
    // linear extrapolation outside the alpha-ellipsoid (modules/poly.py:480-503); rare: one more S tile for the group
    const double beta = sqrt(b2);
    const bool oob = beta > m.alpha;
    if (__any(oob)) {  // (the same in every wave: they hold the same totals)
        double x0[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x0[r] = oob ? (m.alpha * xv[r] + (beta - m.alpha) * k.c_mu[r]) / beta : xv[r];
        __syncthreads();  // (every wave has read RB and XB[0])
#pragma unroll
        for (int r = 0; r < 4; ++r) XB[0][4 * w + r][lane] = x0[r];
        __syncthreads();
        const d4_t j0 = bf_coop128_matvec(k.afS, XB[0], lane);
        double quad0 = 0., lin0 = 0., dotj = 0., j0v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            quad0 += x0[r] * j0[r];
            lin0 += k.c_lin[r] * x0[r];
            j0v[r] = j0[r] + k.c_lin[r];
            dotj += j0v[r] * xm[r];
        }
        quad0 = bf_sum_g(quad0); lin0 = bf_sum_g(lin0); dotj = bf_sum_g(dotj);
        if (g == 0) { RB[0][w][c] = quad0; RB[1][w][c] = lin0; RB[2][w][c] = dotj; }
        __syncthreads();
        quad0 = bf_coop128_total(RB[0], c); lin0 = bf_coop128_total(RB[1], c); dotj = bf_coop128_total(RB[2], c);
        if (oob) {
            const double f0 = (m.c0 + lin0) + 0.5 * quad0;
            f = (beta * f0 - (beta - m.alpha) * m.f_mu) / m.alpha;
            const double coef = (f0 - m.f_mu) / m.alpha - dotj / beta;
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = j0v[r] + coef * (gH[r] / beta);
        }
    }
}

__global__ __launch_bounds__(512) void bf_logp_grad_coop128_kernel(DevModel m, int n, const double *__restrict__ x,
                                                                   double *__restrict__ logp, double *__restrict__ grad) {
    __shared__ double XB[2][Coop128::NS][64];   // B operands of the 32 k-steps: x | x - mu
    __shared__ double RB[4][8][16];             // per-wave partial sums of the 16 points
    Coop128 k;
    bf_coop128_init(k, m);
    const int n_tiles = (n + 15) / 16;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int pt = tile * 16 + k.c;
        double xv[4], gv[4], f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dim = 16 * k.w + 4 * r + k.g;
            xv[r] = (pt < n && dim < m.d) ? x[(size_t)pt * m.d + dim] : 0.;
        }
        bf_coop128_eval(k, m, XB, RB, xv, f, gv);
        if (pt < n) {
            if (k.w == 0 && k.g == 0) logp[pt] = f;
            if (grad) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dim = 16 * k.w + 4 * r + k.g;
                    if (dim < m.d) grad[(size_t)pt * m.d + dim] = gv[r];
                }
            }
        }
    }
}

// CpuLeapfrogIntegrator._step (integration.py:68-95) in the same layout
__global__ __launch_bounds__(512) void bf_leapfrog_coop128_kernel(DevModel m, int n, const double *__restrict__ eps,
                                                                  const double *__restrict__ var, double *__restrict__ q,
                                                                  double *__restrict__ p, double *__restrict__ grad,
                                                                  double *__restrict__ logp, double *__restrict__ energy,
                                                                  double *__restrict__ vel) {
    __shared__ double XB[2][Coop128::NS][64];
    __shared__ double RB[4][8][16];
    Coop128 k;
    bf_coop128_init(k, m);
    const int n_tiles = (n + 15) / 16;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int ch = tile * 16 + k.c;
        const bool live = ch < n;
        const double ep = live ? eps[ch] : 0.;
        const double dt = 0.5 * ep;
        double qn[4], pn[4], vr[4], gn[4], lp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dim = 16 * k.w + 4 * r + k.g;
            const bool ok = live && dim < m.d;
            const size_t idx = (size_t)ch * m.d + dim;
            vr[r] = ok ? var[idx] : 1.;
            const double pp = ok ? p[idx] : 0., gg = ok ? grad[idx] : 0., qq = ok ? q[idx] : 0.;
            pn[r] = pp + dt * gg;              // integration.py:80
            qn[r] = qq + ep * (vr[r] * pn[r]); // :82-85
        }
        bf_coop128_eval(k, m, XB, RB, qn, lp, gn);  // :87
        double kin = 0.;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pn[r] = pn[r] + dt * gn[r];  // :90
            const double v = vr[r] * pn[r];
            kin += pn[r] * v;            // metrics.py:88-91
            vr[r] = v;
        }
        kin = bf_sum_g(kin);
        if (k.g == 0) RB[3][k.w][k.c] = kin;   // (a slot of its own: slower waves may still be reading the evaluation's)
        __syncthreads();
        kin = bf_coop128_total(RB[3], k.c);
        if (live) {
            if (k.w == 0 && k.g == 0) {
                logp[ch] = lp;
                energy[ch] = 0.5 * kin - lp;  // :92-93
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dim = 16 * k.w + 4 * r + k.g;
                if (dim < m.d) {
                    const size_t idx = (size_t)ch * m.d + dim;
                    q[idx] = qn[r];
                    p[idx] = pn[r];
                    grad[idx] = gn[r];
                    if (vel) vel[idx] = vr[r];
                }
            }
        }
    }
}

// CpuLeapfrogIntegrator._step (samplers/hmc_utils/integration.py:68-95), diagonal metric, n chains.
template <int T, bool STAGE, bool PL>
__global__ __launch_bounds__(256) void bf_leapfrog_kernel(DevModel m, int stage_dbl, int n, const double *__restrict__ eps,
                                                          const double *__restrict__ var, double *__restrict__ q,
                                                          double *__restrict__ p, double *__restrict__ grad,
                                                          double *__restrict__ logp, double *__restrict__ energy,
                                                          double *__restrict__ vel) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int E = 4 * T;
    StagedModel sm = bf_stage_model<T, STAGE>(m, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    double *xst = lds + stage_dbl + wave * 16 * 16 * T;
    const int n_tiles = (n + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int ch = tile * 16 + c;
        const bool live = ch < n;
        const double ep = live ? eps[ch] : 0.;
        const double dt = 0.5 * ep;
        double qn[E], pn[E], vr[E], gn[E], lp;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = 4 * e + g;
            const bool ok = live && dim < m.d;
            const size_t idx = (size_t)ch * m.d + dim;
            vr[e] = ok ? var[idx] : 1.;
            const double pp = ok ? p[idx] : 0., gg = ok ? grad[idx] : 0., qq = ok ? q[idx] : 0.;
            pn[e] = pp + dt * gg;              // integration.py:80
            qn[e] = qq + ep * (vr[e] * pn[e]); // :82-85
        }
        bf_eval_w1<T, PL>(m, sm.Sf, sm.Hf, sm.Hdf, sm.pd, 0, qn, lp, gn, lane, xst);  // :87
        double kin = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            pn[e] = pn[e] + dt * gn[e];  // :90
            const double v = vr[e] * pn[e];
            kin += pn[e] * v;            // metrics.py:88-91
            vr[e] = v;
        }
        kin = bf_sum_g(kin);
        if (live) {
            if (g == 0) {
                logp[ch] = lp;
                energy[ch] = 0.5 * kin - lp;  // :92-93
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dim = 4 * e + g;
                if (dim < m.d) {
                    const size_t idx = (size_t)ch * m.d + dim;
                    q[idx] = qn[e];
                    p[idx] = pn[e];
                    grad[idx] = gn[e];
                    if (vel) vel[idx] = vr[e];
                }
            }
        }
    }
}

template <typename K>
static int bf_set_lds(K kern, size_t bytes) {
    if (bytes > 64 * 1024)
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

template <int T, bool STAGE>
static int launch_logp_grad(bfhip_ctx *ctx, int grid, size_t lds, int n, const double *x, int original_space,
                            double *logp, double *grad) {
    const int sd = (int)(bf_stage_bytes(ctx->model) / sizeof(double));
    if (T <= 4 && bf_model_plain(ctx->model)) {  // at d = 128 the plain instantiation schedules worse (more spills)
        auto k = bf_logp_grad_kernel<T, STAGE, (T <= 4)>;
        if (int rc = bf_set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, ctx->stream, ctx->model, sd, n, x, original_space, logp, grad);
    } else {
        auto k = bf_logp_grad_kernel<T, STAGE, false>;
        if (int rc = bf_set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, ctx->stream, ctx->model, sd, n, x, original_space, logp, grad);
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int T, bool STAGE>
static int launch_leapfrog(bfhip_ctx *ctx, int grid, size_t lds, int n, const double *eps, const double *var, double *q,
                           double *p, double *grad, double *logp, double *energy, double *vel) {
    const int sd = (int)(bf_stage_bytes(ctx->model) / sizeof(double));
    if (T <= 4 && bf_model_plain(ctx->model)) {
        auto k = bf_leapfrog_kernel<T, STAGE, (T <= 4)>;
        if (int rc = bf_set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, ctx->stream, ctx->model, sd, n, eps, var, q, p, grad, logp, energy, vel);
    } else {
        auto k = bf_leapfrog_kernel<T, STAGE, false>;
        if (int rc = bf_set_lds(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, ctx->stream, ctx->model, sd, n, eps, var, q, p, grad, logp, energy, vel);
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

static int eval_grid(const bfhip_ctx *ctx, int n) {
    const int n_tiles = (n + 15) / 16;
    int grid = (n_tiles + 3) / 4;
    if (grid > ctx->n_cu * 2) grid = ctx->n_cu * 2;
    return grid;
}

extern "C" int bfhip_logp_grad(bfhip_ctx *ctx, int n, const double *x, int original_space, double *logp, double *grad) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || (n > 0 && (!x || !logp))) return bf_set_error(BFHIP_ERR_ARG, "bfhip_logp_grad: invalid argument");
    if (!ctx->has_model) return bf_set_error(BFHIP_ERR_STATE, "bfhip_logp_grad: no density uploaded");
    if (n == 0) return 0;
    if (ctx->model.pld.on) return bf_pld_logp_grad(ctx, n, x, original_space, logp, grad);
    const int grid = eval_grid(ctx, n);
    const size_t lds = bf_eval_lds_bytes(ctx->model);
    switch (ctx->model.DP / 16) {
    case 1: return launch_logp_grad<1, true>(ctx, grid, lds, n, x, original_space, logp, grad);
    case 2: return launch_logp_grad<2, true>(ctx, grid, lds, n, x, original_space, logp, grad);
    case 4: return launch_logp_grad<4, true>(ctx, grid, lds, n, x, original_space, logp, grad);
    case 8:
        if (bf_model_plain(ctx->model)) {  // (no transform: original_space makes no difference)
            const int n_tiles = (n + 15) / 16;
            hipLaunchKernelGGL(bf_logp_grad_coop128_kernel, dim3(n_tiles < ctx->n_cu ? n_tiles : ctx->n_cu), dim3(512), 0, ctx->stream,
                               ctx->model, n, x, logp, grad);
            BF_HIP_CHECK(hipGetLastError());
            return 0;
        }
        return launch_logp_grad<8, false>(ctx, grid, lds, n, x, original_space, logp, grad);
    }
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "unsupported padded dimension %d", ctx->model.DP);
}

// CpuLeapfrogIntegrator._step around an evaluation that is a launch of its own (the pipeline density: bf_pld_logp_grad):
// one wave per chain, lane = dimension.  first: p += eps/2 g; q += eps var p  (integration.py:80-85).
// second (after logp, grad at the new q): p += eps/2 g'; v = var p; E = p.v / 2 - logp  (:90-93).
template <int E>
__global__ __launch_bounds__(256) void bf_leapfrog_half_kernel(int n, int d, int second, const double *__restrict__ eps,
                                                              const double *__restrict__ var, double *__restrict__ q, double *__restrict__ p,
                                                              const double *__restrict__ grad, const double *__restrict__ logp,
                                                              double *__restrict__ energy, double *__restrict__ velocity_out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const double e = eps[i], dt = 0.5 * e;
    double kin = 0.;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int j = lane + 64 * k;
        if (j < d) {
            const size_t o = (size_t)i * d + j;
            const double pn = p[o] + dt * grad[o];
            p[o] = pn;
            const double v = var[o] * pn;
            if (!second) {
                q[o] = q[o] + e * v;
            } else {
                if (velocity_out) velocity_out[o] = v;
                kin += pn * v;
            }
        }
    }
    if (second) {
        for (int o = 32; o >= 1; o >>= 1) kin += __shfl_xor(kin, o, 64);
        if (lane == 0) energy[i] = 0.5 * kin - logp[i];
    }
}

int bf_pld_logp_grad(bfhip_ctx *ctx, int n, const double *x, int original_space, double *logp, double *grad);

extern "C" int bfhip_leapfrog(bfhip_ctx *ctx, int n, const double *eps, const double *var, double *q, double *p,
                              double *grad, double *logp, double *energy, double *velocity_out) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 0 || (n > 0 && (!eps || !var || !q || !p || !grad || !logp || !energy)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_leapfrog: invalid argument");
    if (!ctx->has_model) return bf_set_error(BFHIP_ERR_STATE, "bfhip_leapfrog: no density uploaded");
    if (n == 0) return 0;
    if (ctx->model.pld.on) {
        // the pipeline density's evaluation is a launch of its own (two contractions shared by a workgroup's points): the step is
        // half step, evaluation, half step -- the same arithmetic as the fused form, three launches
        const int d = ctx->model.d, blocks = (n + 3) / 4;
        if (d <= 64) hipLaunchKernelGGL(bf_leapfrog_half_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, n, d, 0, eps, var, q, p, grad, logp, energy, velocity_out);
        else hipLaunchKernelGGL(bf_leapfrog_half_kernel<2>, dim3(blocks), dim3(256), 0, ctx->stream, n, d, 0, eps, var, q, p, grad, logp, energy, velocity_out);
        BF_HIP_CHECK(hipGetLastError());
        if (int rc = bf_pld_logp_grad(ctx, n, q, 0, logp, grad)) return rc;
        if (d <= 64) hipLaunchKernelGGL(bf_leapfrog_half_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, n, d, 1, eps, var, q, p, grad, logp, energy, velocity_out);
        else hipLaunchKernelGGL(bf_leapfrog_half_kernel<2>, dim3(blocks), dim3(256), 0, ctx->stream, n, d, 1, eps, var, q, p, grad, logp, energy, velocity_out);
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int grid = eval_grid(ctx, n);
    const size_t lds = bf_eval_lds_bytes(ctx->model);
    switch (ctx->model.DP / 16) {
    case 1: return launch_leapfrog<1, true>(ctx, grid, lds, n, eps, var, q, p, grad, logp, energy, velocity_out);
    case 2: return launch_leapfrog<2, true>(ctx, grid, lds, n, eps, var, q, p, grad, logp, energy, velocity_out);
    case 4: return launch_leapfrog<4, true>(ctx, grid, lds, n, eps, var, q, p, grad, logp, energy, velocity_out);
    case 8:
        if (bf_model_plain(ctx->model)) {
            const int n_tiles = (n + 15) / 16;
            hipLaunchKernelGGL(bf_leapfrog_coop128_kernel, dim3(n_tiles < ctx->n_cu ? n_tiles : ctx->n_cu), dim3(512), 0, ctx->stream,
                               ctx->model, n, eps, var, q, p, grad, logp, energy, velocity_out);
            BF_HIP_CHECK(hipGetLastError());
            return 0;
        }
        return launch_leapfrog<8, false>(ctx, grid, lds, n, eps, var, q, p, grad, logp, energy, velocity_out);
    }
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "unsupported padded dimension %d", ctx->model.DP);
}

// transforms/_constraint.pyx:19-215, one thread per coordinate
__global__ void bf_constraint_kernel(DevModel m, int which, long total, const double *__restrict__ x,
                                     double *__restrict__ out, int *__restrict__ bad) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % m.d);
    const double xv = x[idx];
    if (!m.has_transform) {  // core/density.py:93-111
        out[idx] = (which == 0 || which == 3) ? xv : ((which == 1 || which == 4) ? 1. : 0.);
        return;
    }
    const int kind = (int)m.pd[PD_KIND * m.DP + i];
    const double lo = m.pd[PD_LO * m.DP + i], rg = m.pd[PD_RG * m.DP + i];
    double r;
    if (which >= 3) {
        double xo, J, J2;
        bf_to_original(xv, kind, lo, rg, xo, J, J2);
        r = which == 3 ? xo : (which == 4 ? J : J2);
    } else {
        double t = (xv - lo) / rg;
        const bool oob = (kind == 1 && (t <= 0. || t >= 1.)) || (kind == 2 && t <= 0.) || (kind == 3 && t >= 1.);
        if (oob) atomicCAS(bad, 0, i + 1);
        if (which == 0) {
            r = kind == 1 ? log(t / (1. - t)) : (kind == 2 ? log(t) : (kind == 3 ? log(1. - t) : t));
        } else if (which == 1) {
            r = kind == 1 ? 1. / t / (1. - t) : (kind == 2 ? 1. / t : (kind == 3 ? 1. / (t - 1.) : 1.));
            r /= rg;
        } else {
            r = kind == 1 ? (2. * t - 1.) / t / t / (1. - t) / (1. - t)
                          : (kind == 2 ? -1. / t / t : (kind == 3 ? 1. / (t - 1.) / (1. - t) : 0.));
            r /= rg * rg;
        }
    }
    out[idx] = r;
}

extern "C" int bfhip_constraint(bfhip_ctx *ctx, int which, int n, const double *x, double *out, int *bad) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || which < 0 || which > 5 || n < 0 || (n > 0 && (!x || !out)) || (which < 3 && !bad))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_constraint: invalid argument");
    if (!ctx->has_model) return bf_set_error(BFHIP_ERR_STATE, "bfhip_constraint: no density uploaded");
    if (which < 3) BF_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    if (n == 0) return 0;
    const long total = (long)n * ctx->model.d;
    hipLaunchKernelGGL(bf_constraint_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->model,
                       which, total, x, out, bad);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
