// bfhip_tnuts_gen.hip -- tempered NUTS on EVERY density and metric the NUTS kernels run on: cubic configs,
// d up to 128, device-side surrogate input scaling, the Gaussian link, the pipeline density (bfhip_pld.h) and the full-rank metric
// (bfhip_metric.h).  bfhip_tnuts.hip keeps the tuned instantiations of the common surrogate at d <= 64 with the diagonal metric;
// everything it refuses comes here (round 6: closes the "partial" of row f-4).
//
// Reference: TNUTS / TTree (samplers/tnuts.py:15-41) = BaseTHMC.astep (samplers/hmc_utils/base_hmc.py:220-262) around the NUTS tree
// (samplers/nuts.py:21-217) with TCpuLeapfrogIntegrator (samplers/hmc_utils/integration.py:98-222); the target is
// Density.logp_and_grad (core/density.py:724-754) in the sampler's space, the base a quadratic log-density plus log xi.
//
// Layout: as in bfhip_tnuts.hip -- ONE WAVE PER CHAIN, eight waves per workgroup, every evaluation a RENDEZVOUS of the workgroup
// in which the chains' matrix-vector products (S x, H (x - mu), S_b q, H_d (x - mu_d)) run as FP64 MFMA tiles with the chains as
// columns -- generalised: lane l holds dimensions l E .. l E + E - 1 (E = 2 at d = 128), the feature set is decided at run time
// from the uploaded density, the A fragments are streamed from L2 (at d = 128 they do not fit the registers), cubic configs are
// contracted by the chain's own wave from their compact tables, a point outside the bound of a surrogate with cubic configs takes
// a second rendezvous at its projection (modules/poly.py:480-503), and the pipeline density adds its two contractions (three more
// barriers) to the rendezvous, all eight waves sharing them as in the fused NUTS kernel.  The tree logic between two evaluations
// is each wave's own (the reference's recursion, flattened; draws in its post-order), so a chain reproduces the CPU oracle on the
// same xoshiro stream.  Built for coverage and parity; the tuned kernel is the one next door.
#include <cmath>
#include "bfhip_common.h"
#include "bfhip_eval.h"
#include "bfhip_sampler_defs.h"
#include "bfhip_wave.h"
#include "bfhip_oob.h"
#include "bfhip_metric.h"
#include "bfhip_pld.h"
#include "bfhip_tnuts.h"

#define TG_WAVES 8
#define TG_XS 33     // B-operand row: [k = 0 .. 3][chain 0 .. 7] + 1

struct TgLds {   // offsets in doubles
    size_t xb, gb, lsc, flags, pld, total;
};

__host__ __device__ inline TgLds tg_lds_layout(int DP, const PldDev &pl) {
    TgLds L;
    const size_t NS = DP / 4, GS = DP + 1;
    L.xb = 0;
    L.gb = L.xb + 4 * NS * TG_XS;
    L.lsc = L.gb + 4 * 8 * GS;
    L.flags = L.lsc + (size_t)TG_WAVES * TN_MAXL * TS_N;
    L.pld = (L.flags + 2 + 1) & ~(size_t)1;
    L.total = L.pld + (pl.on ? pld_lds_doubles(DP, pl.MP, pl.PP, pl.KS2, pl.n_ent, PLD_XS8) : 0);
    return L;
}

struct TgBase { const double *S, *lin; double c0, logxi; };

template <int W, bool FULLM>
__global__ __launch_bounds__(64 * TG_WAVES) void bf_tnuts_gen_kernel(DevModel m, TnutsArgs a, double *mat_all) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int DP = 16 * W, NS = 4 * W, E = W == 8 ? 2 : 1, GS = DP + 1;
    const int d = a.d;
    const PldDev &pl = m.pld;
    const TgLds LL = tg_lds_layout(DP, pl);
    double *XB = lds + LL.xb;            // [4][NS][TG_XS]   x | x - mu | q (base) | x_o - mu_decay
    double *GB = lds + LL.gb;            // [4][8][GS]       S x | H (x - mu) | S_b q | H_d (x_o - mu_d)
    double *LSC = lds + LL.lsc;          // [8][TN_MAXL][TS_N]
    int *flags = (int *)(lds + LL.flags);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chain = blockIdx.x * a.cpg + w;
    const bool real = w < a.cpg && chain < a.n_chain;
    const bool f_quad = m.has_quad != 0, f_bound = m.use_bound != 0, f_decay = m.use_decay != 0, f_tr = m.has_transform != 0;
    const bool f_su = m.has_su != 0, f_cubic = m.has_cubic != 0, f_link = m.has_link != 0, f_pld = pl.on != 0;
    PldLds PL;
    if (f_pld) {
        PL = pld_lds(lds + LL.pld, DP, pl, 8);
        pld_stage(pl, PL, DP, threadIdx.x, 64 * TG_WAVES);
    }
    if (threadIdx.x < 4) flags[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < (int)(LL.lsc - LL.xb); i += 64 * TG_WAVES) XB[i] = 0.;   // (XB and GB: the columns without a chain stay zero)
    __syncthreads();
    double *lsw = LSC + w * (TN_MAXL * TS_N);
    bool in[E];
    double c_lin[E], c_mu[E], c_smu[E], b_lin[E], c_dmu[E], c_lo[E], c_rg[E], c_sulo[E], c_sudf[E];
    int c_kind[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int dim = lane * E + e;
        in[e] = dim < d;
        const bool tab = dim < DP;
        c_lin[e] = tab ? m.pd[PD_LIN * DP + dim] : 0.;
        c_mu[e] = tab ? m.pd[PD_MU * DP + dim] : 0.;
        c_smu[e] = tab ? m.pd[PD_SMU * DP + dim] : 0.;
        c_dmu[e] = (f_decay && tab) ? m.pd[PD_DMU * DP + dim] : 0.;
        c_kind[e] = (f_tr && in[e]) ? (int)m.pd[PD_KIND * DP + dim] : 0;
        c_lo[e] = (f_tr && in[e]) ? m.pd[PD_LO * DP + dim] : 0.;
        c_rg[e] = (f_tr && in[e]) ? m.pd[PD_RG * DP + dim] : 1.;
        c_sulo[e] = (f_su && in[e]) ? m.pd[PD_SU_LO * DP + dim] : 0.;
        c_sudf[e] = (f_su && in[e]) ? m.pd[PD_SU_DIFF * DP + dim] : 1.;
        b_lin[e] = in[e] ? a.base_lin[dim] : 0.;
    }
    const size_t msz = (size_t)d * d;
    double *matp = FULLM ? mat_all + (size_t)(real ? chain : 0) * BF_MAT_N * msz : nullptr;

    // ---- one rendezvous of the workgroup: the matrix-vector products of every chain's point ----
    // jobs (matrix mi, row tile t): j = mi W + t, dealt as j = w, w + 8, ...; two v_mfma_f64_4x4x4_4b per k-step (columns 0-3 and
    // 4-7: bfhip_tnuts.hip has the operand maps)
    constexpr int JPW = (4 * W + TG_WAVES - 1) / TG_WAVES;
    int n_x = 0;
    auto rdv_matvec = [&](bool active, const double (&x)[E], const double (&xm)[E], const double (&qb)[E], const double (&xd)[E],
                          double (&sx)[E], double (&hv)[E], double (&bx)[E], double (&dgr)[E]) -> bool {
        const int par = n_x & 1;
        n_x += 1;
        if (active) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dim = lane * E + e;
                if (dim < DP) {
                    const int xi = (dim >> 2) * TG_XS + 8 * (dim & 3) + w;
                    XB[0 * NS * TG_XS + xi] = x[e];
                    XB[1 * NS * TG_XS + xi] = xm[e];
                    XB[2 * NS * TG_XS + xi] = qb[e];
                    XB[3 * NS * TG_XS + xi] = xd[e];
                }
            }
            if (lane == 0) flags[par] = 1;
        }
        __syncthreads();  // R1
        const bool any = rfl(flags[par]) != 0;
        if (threadIdx.x == 0) flags[par ^ 1] = 0;
        if (any) {
            double lo[JPW], hi[JPW];
            const double *Af[JPW];
            const double *Bq[JPW];
            bool on[JPW], dense[JPW];
            int jt[JPW], jm[JPW];
#pragma unroll
            for (int j = 0; j < JPW; ++j) {
                const int job = w + TG_WAVES * j;
                jm[j] = job / W;
                jt[j] = job % W;
                const int mi = jm[j];
                on[j] = job < 4 * W && ((mi == 0 && f_quad) || (mi == 1 && f_bound) || mi == 2 || (mi == 3 && f_decay));
                dense[j] = mi == 2;
                Af[j] = (mi == 0 ? m.Sf : (mi == 1 ? m.Hf : m.Hdf)) + (size_t)jt[j] * NS * 64 + lane;
                Bq[j] = XB + (size_t)(mi < 4 ? mi : 0) * NS * TG_XS + 8 * (lane >> 4) + (lane & 3);
                lo[j] = 0.;
                hi[j] = 0.;
            }
            for (int s0 = 0; s0 < NS; s0 += 4) {
                double av[JPW][4], bl[JPW][4], bh[JPW][4];
#pragma unroll
                for (int j = 0; j < JPW; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int s = s0 + q;
                        double v = 0.;
                        if (on[j]) {
                            if (dense[j]) {
                                const int row = 16 * jt[j] + (lane & 15), col = 4 * s + (lane >> 4);
                                v = (row < d && col < d) ? a.base_S[(size_t)row * d + col] : 0.;
                            } else {
                                v = Af[j][(size_t)s * 64];
                            }
                        }
                        av[j][q] = v;
                        bl[j][q] = Bq[j][s * TG_XS];
                        bh[j][q] = Bq[j][s * TG_XS + 4];
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < JPW; ++j) {
                        lo[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[j][q], bl[j][q], lo[j], 0, 0, 0);
                        hi[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[j][q], bh[j][q], hi[j], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int j = 0; j < JPW; ++j)
                if (w + TG_WAVES * j < 4 * W) {
                    const int col = lane & 3, row = 16 * jt[j] + 4 * ((lane >> 2) & 3) + (lane >> 4);
                    GB[((size_t)jm[j] * 8 + col) * GS + row] = lo[j];
                    GB[((size_t)jm[j] * 8 + 4 + col) * GS + row] = hi[j];
                }
        }
        __syncthreads();  // R2
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = lane * E + e;
            const bool rd = dim < DP;
            sx[e] = rd ? GB[((size_t)0 * 8 + w) * GS + dim] : 0.;
            hv[e] = rd ? GB[((size_t)1 * 8 + w) * GS + dim] : 0.;
            bx[e] = rd ? GB[((size_t)2 * 8 + w) * GS + dim] : 0.;
            dgr[e] = rd ? GB[((size_t)3 * 8 + w) * GS + dim] : 0.;
        }
        return any;
    };
    // the pipeline density's part of a rendezvous: every wave takes its share of the two contractions (three barriers)
    auto rdv_pld = [&]() {
        __syncthreads();  // P1: the monomials of every evaluating chain
        pld_gemm1_q8(pl, PL, m.alpha, w, TG_WAVES, lane);
        __syncthreads();  // P2: residuals
        pld_gemm2_q8(pl, PL, w, TG_WAVES, lane);
        __syncthreads();  // P3: W = C'^T r
    };

    // element k of a vector of this chain (wave-uniform k / per-lane k)
    auto xu = [&](const double (&v)[E], int k) { return readlane_f64((E > 1 && (k % E)) ? v[E - 1] : v[0], k / E); };
    auto xl = [&](const double (&v)[E], int k) {
        const double a0 = __shfl(v[0], k / E, 64);
        if constexpr (E > 1) { const double a1 = __shfl(v[E - 1], k / E, 64); return (k % E) ? a1 : a0; }
        return a0;
    };
    // cubic configs (modules/_poly.pyx:49-137) of this chain's point: gradient added to gn, value returned (bfhip_sampler.hip has
    // the layout: lane (jl, kq) accumulates output index j = jb + jl over k = kq, kq + 4, ...; compact tables from L2)
    auto cubic_add = [&](const double (&xev)[E], double (&gn)[E]) -> double {
        const int jl = lane & 15, kq = lane >> 4;
        auto fetch = [&](const int *pos, int jb, double val) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dim = lane * E + e;
                const int pj = dim < m.DP ? pos[dim] : -1;
                const bool mine = pj >= jb && pj < jb + 16;
                const double gv = __shfl(val, mine ? pj - jb : 0, 64);
                if (mine) gn[e] += gv;
            }
        };
        double fsum = 0.;
        for (int jb = 0; jb < m.n2; jb += 16) {   // cubic-2: f = sum_j x_j^2 v1_j, v1 = A x; df/dx_j = 2 x_j v1_j + (A^T x^2)_j
            const int j = jb + jl;
            const bool onj = j < m.n2;
            double v1 = 0., v2 = 0.;
            for (int kk = 0; kk < m.n2; kk += 4) {
                const int k = kk + kq;
                const bool ok = onj && k < m.n2;
                const double xk = xl(xev, ok ? m.mask2[k] : 0);
                v1 += (ok ? m.A2t[k * m.n2 + j] : 0.) * xk;
                v2 += (ok ? m.A2[k * m.n2 + j] : 0.) * (xk * xk);
            }
            v1 = swap32_add_f64(swap16_add_f64(v1));
            v2 = swap32_add_f64(swap16_add_f64(v2));
            const double xj = xl(xev, onj ? m.mask2[j] : 0);
            if (onj && kq == 0) fsum += xj * xj * v1;
            fetch(m.pos2, jb, 2. * xj * v1 + v2);
        }
        for (int jb = 0; jb < m.n3; jb += 16) {   // cubic-3: df/dx_j = 1/2 sum_{k,l} T[j,k,l] x_k x_l, f = x . grad / 3
            const int j = jb + jl;
            const bool onj = j < m.n3;
            double sacc = 0.;
            for (int kk = 0; kk < m.n3; kk += 4) {
                const int k = kk + kq;
                const bool ok = onj && k < m.n3;
                double t = 0.;
                const double *Tk = m.T3t + (size_t)(ok ? k : 0) * m.n3 * m.n3 + (ok ? j : 0);
                for (int l = 0; l < m.n3; ++l) t += (ok ? Tk[(size_t)l * m.n3] : 0.) * xu(xev, rfl(m.mask3[l]));
                sacc += t * xl(xev, ok ? m.mask3[k] : 0);
            }
            sacc = swap32_add_f64(swap16_add_f64(sacc));
            const double xj = xl(xev, onj ? m.mask3[j] : 0);
            if (onj && kq == 0) fsum += xj * (0.5 * sacc) * (1. / 3.);
            fetch(m.pos3, jb, 0.5 * sacc);
        }
        return wave_sum(fsum);
    };

    // phi, dphi, psi, dpsi at q: integration.py:180-181 / base_hmc.py:227-231; the target as Density.logp_and_grad evaluates it
    // (core/density.py:724-754: transform, surrogate scaling, polynomial with its bound, chain rule, link / chi-square / prior,
    // decay, log-Jacobian), expression for expression as bf_eval_w1 (bfhip_eval.h) and the fused NUTS kernel's phase P
    auto potentials = [&](const double (&q)[E], double &phi, double (&dphi)[E], double &psi, double (&dpsi)[E]) {
        double xo[E], jac[E], gj[E], xs[E], xm[E], xd[E], qb[E];
        double logdet_l = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            xo[e] = in[e] ? q[e] : 0.;
            jac[e] = 1.;
            gj[e] = 0.;
            if (f_tr && in[e]) {
                double J, J2;
                bf_to_original(q[e], c_kind[e], c_lo[e], c_rg[e], xo[e], J, J2);
                logdet_l += log(fabs(J));
                jac[e] = J;
                gj[e] = J2 / J;
            }
            xs[e] = (f_su && in[e]) ? (xo[e] - c_sulo[e]) / c_sudf[e] : xo[e];
            xm[e] = in[e] ? xs[e] - c_mu[e] : 0.;
            xd[e] = (f_decay && in[e]) ? xo[e] - c_dmu[e] : 0.;
            qb[e] = in[e] ? q[e] : 0.;
        }
        double sx[E], hv[E], bx[E], dgr[E], gn[E];
        (void)rdv_matvec(true, xs, xm, qb, xd, sx, hv, bx, dgr);
        double f;
        double r_base = 0., r_bd2 = 0.;
        if (f_pld) {
            // the bound is decided first, so a point outside the ellipsoid is evaluated once, at its projection
            double r2[4] = {0., 0., 0., 0.};
#pragma unroll
            for (int e = 0; e < E; ++e) {
                r2[0] += xm[e] * hv[e];
                r2[1] += xd[e] * dgr[e];
                r2[2] += in[e] ? __builtin_fma(0.5 * q[e], bx[e], b_lin[e] * q[e]) : 0.;
            }
            r2[3] = logdet_l;
            wave_sum_n<4>(r2);
            const double r_b2 = r2[0];
            r_bd2 = r2[1];
            r_base = r2[2];
            logdet_l = r2[3];
            double beta_o = 0.;
            if (f_bound && !(r_b2 < m.alpha * m.alpha * (1. - 1e-12))) {   // modules/poly.py:467-469
                const double b = rfl(sqrt(r_b2));
                if (b > m.alpha) beta_o = b;
            }
            {   // pld_point (bfhip_pld.h) for E elements per lane
                double *xe = PL.XE + w * (DP + 2);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int dim = lane * E + e;
                    const double x_ev = beta_o > 0. ? (m.alpha * xs[e] + (beta_o - m.alpha) * c_mu[e]) / beta_o : xs[e];   // :482
                    if (dim < DP) xe[dim] = in[e] ? x_ev : 0.;
                }
                if (lane == 0) { xe[DP] = 1.; xe[DP + 1] = 0.; PL.CH[w] = beta_o; }
                for (int p0 = 0; p0 < pl.PP; p0 += 64) {
                    const int p = p0 + lane;
                    if (p < pl.PP) {
                        const unsigned mo = PL.MONO[p];
                        PL.PHI[(p >> 2) * PL.XS + w + PL.CW * (p & 3)] = (xe[mo & 255u] * xe[(mo >> 8) & 255u]) * xe[(mo >> 16) & 255u];
                    }
                }
            }
            rdv_pld();
            double s2[2];
            pld_sums(pl, PL, w, lane, TG_WAVES, s2[0], s2[1]);
            wave_sum_n<2>(s2);
            double gj0[E], dj = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dim = lane * E + e;
                gj0[e] = dim < DP ? pld_grad(pl, PL, DP, w, dim) : 0.;   // (J_0^T r)_dim
                dj += gj0[e] * xm[e];
            }
            if (beta_o > 0.) {   // (compressed outputs: the tails of Q^T f_mu' and Q^T y' as scalars, bfhip_pipeline_upload)
                const double b = (beta_o - m.alpha) / m.alpha;
                s2[0] += b * (b * pl.k_ff + 2. * pl.k_fy);
                s2[1] += b * pl.k_ff + pl.k_fy;
                const double r_dotj = wave_sum(dj);   // modules/poly.py:494-496, contracted with r
#pragma unroll
                for (int e = 0; e < E; ++e) gj0[e] += (s2[1] / m.alpha - r_dotj / beta_o) * (hv[e] / beta_o);
            }
            f = pl.logp0 - 0.5 * s2[0];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                gn[e] = -gj0[e];                          // density.py:552-560
                if (f_su) gn[e] = gn[e] / c_sudf[e];      // module.py:226
                gn[e] = gn[e] * jac[e];                   // density.py:558
            }
            if (pl.has_prior) {
                double pr = 0.;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int dim = lane * E + e;
                    const double dx = in[e] ? xo[e] - pl.prior_mu[dim] : 0., pp = in[e] ? pl.prior_prec[dim] : 0.;
                    pr += pp * dx * dx;
                    gn[e] += -(pp * dx) * jac[e];
                }
                f += pl.prior_c0 - 0.5 * wave_sum(pr);
            }
        } else {
            double r7[7] = {0., 0., 0., 0., 0., 0., 0.};
#pragma unroll
            for (int e = 0; e < E; ++e) {
                gn[e] = sx[e] + c_lin[e];
                const double sv = sx[e] - c_smu[e], gmu = c_smu[e] + c_lin[e];
                r7[0] += in[e] ? __builtin_fma(0.5 * xs[e], sx[e], c_lin[e] * xs[e]) : 0.;
                r7[1] += xm[e] * hv[e];
                r7[2] += in[e] ? __builtin_fma(0.5 * q[e], bx[e], b_lin[e] * q[e]) : 0.;
                r7[3] += xm[e] * gmu;
                r7[4] += xm[e] * sv;
                r7[6] += xd[e] * dgr[e];
            }
            r7[5] = logdet_l;
            wave_sum_n<7>(r7);
            r_base = r7[2];
            logdet_l = r7[5];
            r_bd2 = r7[6];
            double fcub = 0.;
            if (f_cubic) fcub = cubic_add(xs, gn);
            f = (m.c0 + r7[0]) + fcub;
            if (f_bound) {
                const double beta = rfl(sqrt(r7[1]));
                if (beta > m.alpha) {
                    if (!f_cubic) {   // S x_0 follows from S x by linearity (bfhip_oob.h)
                        const BfOob o = bf_oob_scalars(m.alpha, m.inv_alpha, m.f_mu, m.f_poly_mu, beta, r7[3], r7[4]);
                        f = o.f;
#pragma unroll
                        for (int e = 0; e < E; ++e) gn[e] = bf_oob_grad(o, c_smu[e] + c_lin[e], sx[e] - c_smu[e], hv[e]);
                    } else {          // cubic configs are not linear in x: a second rendezvous at the projected point (modules/poly.py:480-503)
                        double x0[E], sx0[E], t1[E], t2[E], t3[E], j0[E];
#pragma unroll
                        for (int e = 0; e < E; ++e) x0[e] = in[e] ? (m.alpha * xs[e] + (beta - m.alpha) * c_mu[e]) / beta : 0.;
                        (void)rdv_matvec(true, x0, xm, qb, xd, sx0, t1, t2, t3);
                        double r2[2] = {0., 0.};
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            j0[e] = sx0[e] + c_lin[e];
                            r2[0] += in[e] ? __builtin_fma(0.5 * x0[e], sx0[e], c_lin[e] * x0[e]) : 0.;
                        }
                        const double fcub0 = cubic_add(x0, j0);
#pragma unroll
                        for (int e = 0; e < E; ++e) r2[1] += j0[e] * xm[e];
                        wave_sum_n<2>(r2);
                        const double f0 = (m.c0 + r2[0]) + fcub0;
                        f = (beta * f0 - (beta - m.alpha) * m.f_mu) / m.alpha;
                        const double coef = (f0 - m.f_mu) / m.alpha - r2[1] / beta;
#pragma unroll
                        for (int e = 0; e < E; ++e) gn[e] = j0[e] + coef * (hv[e] / beta);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (f_su) gn[e] = gn[e] / c_sudf[e];   // module.py:226
                gn[e] = gn[e] * jac[e];                // density.py:558
            }
            if (f_link) {  // the next module of the pipeline (density.py:552-560)
                const double r = f - m.link_y, dl = -(m.link_prec * r);
                f = m.link_logp0 - 0.5 * (r * (m.link_prec * r));
#pragma unroll
                for (int e = 0; e < E; ++e) gn[e] = dl * gn[e];
            }
        }
        if (f_decay) {  // density.py:740-746
            f -= m.decay_gamma * bf_clip0(r_bd2 - m.decay_alpha2);
            if (r_bd2 > m.decay_alpha2) {
#pragma unroll
                for (int e = 0; e < E; ++e) gn[e] -= 2. * m.decay_gamma * dgr[e];
            }
        }
        if (f_tr) {   // density.py:747-750
            f += logdet_l;
#pragma unroll
            for (int e = 0; e < E; ++e) gn[e] += gj[e];
        }
        phi = rfl(-f);
        psi = rfl(-((a.base_c0 + r_base) + a.logxi));
#pragma unroll
        for (int e = 0; e < E; ++e) {
            dphi[e] = in[e] ? -gn[e] : 0.;
            dpsi[e] = in[e] ? -(bx[e] + b_lin[e]) : 0.;
        }
    };

    if (real) {
        // ---- chain state ----
        double *scp = a.sc + (size_t)chain * BFHIP_SC_N;
        double *vecp = a.vec + (size_t)chain * BFHIP_VEC_N * d;
        double *sb = a.scratch + (size_t)chain * (4 * TN_MAXL) * DP + lane * E;
        uint64_t rs[4];
        for (int k = 0; k < 4; ++k) rs[k] = a.rng[(size_t)chain * 4 + k];
        double log_step = scp[BFHIP_SC_LOG_STEP], log_bar = scp[BFHIP_SC_LOG_BAR], hbar = scp[BFHIP_SC_HBAR];
        const double smu = scp[BFHIP_SC_MU];
        double count = scp[BFHIP_SC_COUNT];
        double fg_n = scp[BFHIP_SC_FG_N], bg_n = scp[BFHIP_SC_BG_N], n_samples = scp[BFHIP_SC_N_SAMPLES];
        double prev_upd = scp[BFHIP_SC_PREV_UPDATE], adapt_window = scp[BFHIP_SC_ADAPT_WINDOW];
        int i_iter = (int)scp[BFHIP_SC_I_ITER], err = (int)scp[BFHIP_SC_ERROR];
        double qc[E], var[E];
        auto load_vec = [&](int field, double (&v)[E], double pad) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = in[e] ? vecp[field * d + lane * E + e] : pad;
        };
        auto store_vec = [&](int field, const double (&v)[E]) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (in[e]) vecp[field * d + lane * E + e] = v[e];
        };
        load_vec(BFHIP_VEC_Q, qc, 0.);
        load_vec(BFHIP_VEC_VAR, var, 1.);
        double u_cur = rfl(a.tu[chain]);
        unsigned long long nlf = 0;
        auto uni = [&]() { return bf_u01(bf_xoshiro_next(rs)); };
        auto logbern = [&](double l) -> bool {  // nuts.py:200-203
            if (l != l) err = 2;
            return log(uni()) < l;
        };
        const bool lane_ok = lane * E < DP;   // (lanes beyond the padded dimension hold zeros and own no slot words)
        auto ldv = [&](int slot, double (&v)[E]) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = lane_ok ? sb[(size_t)slot * DP + e] : 0.;
        };
        auto stv = [&](int slot, const double (&v)[E]) {
            if (lane_ok) {
#pragma unroll
                for (int e = 0; e < E; ++e) sb[(size_t)slot * DP + e] = v[e];
            }
        };
        // velocity of a momentum: metrics.py:88-91 (diagonal), :113-115 (full rank)
        auto vel = [&](const double (&p)[E], double (&out)[E]) {
            if constexpr (FULLM) {
                bf_velocity_full<E>(matp + BF_MAT_COV * msz, p, out, d, lane);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) out[e] = var[e] * p[e];
            }
        };

        // one tempered leapfrog step from (q, p, u, vt): integration.py:153-222
        struct TS { double q[E], p[E], v[E]; double u, vt, weight, energy, logp; };
        auto finish_state = [&](TS &s, double phi, double psi) {
            double k = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e) k += s.p[e] * s.v[e];
            const double kin = tn_wsum(k);
            const double ope = 1 + exp(-s.u), beta = 1 / ope, pot = s.u + 2 * log(ope);
            s.energy = rfl((beta * phi + (1 - beta) * psi + pot) + (0.5 * kin + s.vt * s.vt / 2));
            s.logp = rfl(-phi);
            s.weight = rfl(phi - psi);
        };
        auto t_step = [&](const TS &s0, double eps) -> TS {
            TS s = s0;
            const double dt = 0.5 * eps;
            double phi, dphi[E], psi, dpsi[E];
            s.u = rfl(s.u + s.vt * dt);
#pragma unroll
            for (int e = 0; e < E; ++e) s.q[e] += dt * s.v[e];
            potentials(s.q, phi, dphi, psi, dpsi);
            const double ex = exp(-s.u), beta = 1 / (1 + ex), dbeta = ex * beta * beta, dU = (1 - ex) * beta;
            s.vt = rfl(s.vt + -(dbeta * (phi - psi) + dU) * eps);
#pragma unroll
            for (int e = 0; e < E; ++e) s.p[e] += eps * -(beta * dphi[e] + (1 - beta) * dpsi[e]);
            s.u = rfl(s.u + s.vt * dt);
            vel(s.p, s.v);
#pragma unroll
            for (int e = 0; e < E; ++e) s.q[e] += dt * s.v[e];
            potentials(s.q, phi, dphi, psi, dpsi);
            finish_state(s, phi, psi);
            return s;
        };
        auto dot6 = [&](const double *a0, const double *b0, const double *a1, const double *b1, const double *a2, const double *b2,
                        const double *a3, const double *b3, const double *a4, const double *b4, const double *a5, const double *b5) -> bool {
            double r6[6] = {0., 0., 0., 0., 0., 0.};
#pragma unroll
            for (int e = 0; e < E; ++e) {
                r6[0] += a0[e] * b0[e]; r6[1] += a1[e] * b1[e]; r6[2] += a2[e] * b2[e];
                r6[3] += a3[e] * b3[e]; r6[4] += a4[e] * b4[e]; r6[5] += a5[e] * b5[e];
            }
            wave_sum_n<6>(r6);
            return (r6[0] <= 0.) || (r6[1] <= 0.) || (r6[2] <= 0.) || (r6[3] <= 0.) || (r6[4] <= 0.) || (r6[5] <= 0.);
        };

        while (i_iter < a.iter_end && err == 0) {
            const bool warm = i_iter < a.cfg.n_warmup;
            // ---- BaseTHMC.astep: base_hmc.py:233-262 ----
            TS start;
            {   // p0 = metric.random: one xoshiro draw keys the SplitMix64 stream of the d normals (as in the other kernels)
                const uint64_t K = bf_xoshiro_next(rs);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int dim = lane * E + e;
                    const uint64_t P = (uint64_t)(dim >> 1);
                    const double u1 = bf_u01_open0(bf_mix64(K + (2 * P + 1) * BF_GOLDEN)), u2 = bf_u01(bf_mix64(K + (2 * P + 2) * BF_GOLDEN));
                    const double rad = sqrt(-2. * log(u1));
                    double sn, cs;
                    sincospi(2. * u2, &sn, &cs);
                    const double z = (dim & 1) ? rad * sn : rad * cs;
                    if constexpr (FULLM) start.p[e] = in[e] ? z : 0.;
                    else start.p[e] = in[e] ? (1. / sqrt(var[e])) * z : 0.;
                    start.q[e] = qc[e];
                }
                if constexpr (FULLM) bf_solve_lt<E>(matp + BF_MAT_CHOL_ROWS * msz, start.p, d, lane);  // metrics.py:123-127
            }
            {   // v0 = rng.normal(0, 1): a stream of its own, first (cosine) element
                const uint64_t K = bf_xoshiro_next(rs);
                const double u1 = bf_u01_open0(bf_mix64(K + BF_GOLDEN)), u2 = bf_u01(bf_mix64(K + 2 * BF_GOLDEN));
                double sn, cs;
                sincospi(2. * u2, &sn, &cs);
                start.vt = rfl(sqrt(-2. * log(u1)) * cs);
            }
            start.u = u_cur;
            vel(start.p, start.v);
            {
                double phi, dphi[E], psi, dpsi[E];
                potentials(start.q, phi, dphi, psi, dpsi);
                finish_state(start, phi, psi);
            }
            if (!(fabs(start.energy) <= 1.7976931348623157e308)) { err = 1; break; }
            const double eps0 = rfl(exp(warm ? log_step : log_bar));
            // ---- Tree.__init__: nuts.py:24-43 ----
            TS left = start, right = start;
            double prop_q[E], p_sum[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { prop_q[e] = start.q[e]; p_sum[e] = start.p[e]; }
            double prop_u = start.u, prop_w = start.weight, prop_E = start.energy, prop_logp = start.logp;
            double log_size = 0., accept_sum = 0., max_de = 0.;
            int depth = 0, n_prop = 0, diverging = 0, turning = 0;
            for (int it = 0; it < a.cfg.max_treedepth && err == 0; ++it) {
                const int dir = logbern(-0.6931471805599453094) ? 1 : -1;  // nuts.py:210
                const double eps = dir > 0 ? eps0 : -eps0;
                const TS old_left = left, old_right = right;
                // ---- _build_subtree(edge, depth, eps), recursion flattened: leaf i merges upwards while bit `lev` of i is set ----
                TS cur = dir > 0 ? right : left;
                double T_lp[E], T_ps[E], T_pq[E], L0_lp[E], L0_rp[E], L0_ps[E], L0_pq[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { T_lp[e] = T_ps[e] = T_pq[e] = L0_lp[e] = L0_rp[e] = L0_ps[e] = L0_pq[e] = 0.; }
                double T_pu = 0., T_pw = 0., T_pE = 0., T_plogp = 0., T_ls = 0., T_acc = 0.;
                double sub_acc = 0.;
                long sub_n = 0;
                bool done = false;
                const int n_leaf = 1 << depth;
                for (int i_leaf = 0; i_leaf < n_leaf && !done; ++i_leaf) {
                    // ---- _single_step: nuts.py:105-132 ----
                    const TS nxt = t_step(cur, eps);
                    nlf += 1;
                    sub_n += 1;
                    double dE = rfl(nxt.energy - start.energy);
                    if (dE != dE) dE = INFINITY;
                    if (fabs(dE) > fabs(max_de)) max_de = dE;
                    if (!(fabs(dE) < a.cfg.max_change)) {
                        diverging = 1;
                        // the stub subtree: ancestors still add their left halves' accept sums (nuts.py:173)
                        for (int al = 0; al < depth; ++al)
                            if ((i_leaf >> al) & 1) sub_acc = rfl(sub_acc + lsw[al * TS_N + TS_ACC]);
                        done = true;
                        break;
                    }
                    cur = nxt;
#pragma unroll
                    for (int e = 0; e < E; ++e) { T_lp[e] = nxt.p[e]; T_ps[e] = nxt.p[e]; T_pq[e] = nxt.q[e]; }
                    T_pu = nxt.u; T_pw = nxt.weight; T_pE = nxt.energy; T_plogp = nxt.logp;
                    T_ls = -dE;
                    { const double pa = rfl(exp(-dE)); T_acc = pa > 1. ? 1. : pa; }
                    int lev = 0;
                    bool abort = false;
                    while (lev < depth && ((i_leaf >> lev) & 1)) {
                        // ---- merge with the waiting left sibling of this level: nuts.py:146-178 ----
                        double A_lp[E], A_rp[E], A_ps[E], A_pq[E];
                        if (lev == 0) {
#pragma unroll
                            for (int e = 0; e < E; ++e) { A_lp[e] = L0_lp[e]; A_rp[e] = L0_rp[e]; A_ps[e] = L0_ps[e]; A_pq[e] = L0_pq[e]; }
                        } else {
                            ldv(4 * lev + 0, A_lp); ldv(4 * lev + 1, A_rp); ldv(4 * lev + 2, A_ps); ldv(4 * lev + 3, A_pq);
                        }
                        const double *ls = lsw + lev * TS_N;
                        double psum[E];
#pragma unroll
                        for (int e = 0; e < E; ++e) psum[e] = A_ps[e] + T_ps[e];
                        bool turn;
                        if (lev >= 1) {  // with the sub-span checks for depth > 1 (nuts.py:154-161): six sums, one reduction
                            double A_lv[E], A_rv[E], T_lv[E], ps1[E], ps2[E];
                            if constexpr (FULLM) bf_velocity_full3<E>(matp + BF_MAT_COV * msz, A_lp, A_rp, T_lp, A_lv, A_rv, T_lv, d, lane);
                            else { vel(A_lp, A_lv); vel(A_rp, A_rv); vel(T_lp, T_lv); }
#pragma unroll
                            for (int e = 0; e < E; ++e) { ps1[e] = A_ps[e] + T_lp[e]; ps2[e] = A_rp[e] + T_ps[e]; }
                            turn = dot6(psum, A_lv, psum, cur.v, ps1, A_lv, ps1, T_lv, ps2, A_rv, ps2, cur.v);
                        } else {
                            double A_lv[E];
                            vel(A_lp, A_lv);
                            double r2[2] = {0., 0.};
#pragma unroll
                            for (int e = 0; e < E; ++e) { r2[0] += psum[e] * A_lv[e]; r2[1] += psum[e] * cur.v[e]; }
                            wave_sum_n<2>(r2);
                            turn = (r2[0] <= 0.) || (r2[1] <= 0.);
                        }
                        const double acc_l = rfl(ls[TS_ACC]), ls_l = rfl(ls[TS_LS]);
                        const double ls_new = rfl(tn_logaddexp(ls_l, T_ls));
                        const bool take2 = logbern(T_ls - ls_new);  // :164 (drawn even when this merge turns)
                        T_acc = rfl(acc_l + T_acc);
                        if (turn) {
                            for (int al = lev + 1; al < depth; ++al)
                                if ((i_leaf >> al) & 1) T_acc = rfl(T_acc + lsw[al * TS_N + TS_ACC]);
                            abort = true;
                            turning = 1;
                            break;
                        }
                        if (!take2) {
#pragma unroll
                            for (int e = 0; e < E; ++e) T_pq[e] = A_pq[e];
                            T_pE = rfl(ls[TS_E]); T_plogp = rfl(ls[TS_LOGP]); T_pu = rfl(ls[TS_U]); T_pw = rfl(ls[TS_W]);
                        }
                        T_ls = ls_new;
#pragma unroll
                        for (int e = 0; e < E; ++e) { T_ps[e] = psum[e]; T_lp[e] = A_lp[e]; }
                        lev += 1;
                    }
                    if (abort) { sub_acc = T_acc; done = true; break; }
                    if (lev < depth) {
                        // wait for the right sibling
                        if (lev == 0) {
#pragma unroll
                            for (int e = 0; e < E; ++e) { L0_lp[e] = T_lp[e]; L0_rp[e] = cur.p[e]; L0_ps[e] = T_ps[e]; L0_pq[e] = T_pq[e]; }
                        } else {
                            stv(4 * lev + 0, T_lp); stv(4 * lev + 1, cur.p); stv(4 * lev + 2, T_ps); stv(4 * lev + 3, T_pq);
                        }
                        double *ls = lsw + lev * TS_N;
                        if (lane == 0) { ls[TS_LS] = T_ls; ls[TS_ACC] = T_acc; ls[TS_E] = T_pE; ls[TS_LOGP] = T_plogp; ls[TS_U] = T_pu; ls[TS_W] = T_pw; }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    } else {
                        sub_acc = T_acc;  // the whole subtree of this doubling is complete
                    }
                }
                depth += 1;
                accept_sum = rfl(accept_sum + sub_acc);
                n_prop += (int)sub_n;
                if (err) break;
                if (diverging || turning) break;   // Tree.extend returns before touching the ends' p_sum (nuts.py:71-73)
                // ---- Tree.extend after a complete subtree: nuts.py:75-103 ----
                if (dir > 0) right = cur; else left = cur;
                if (logbern(T_ls - log_size)) {
#pragma unroll
                    for (int e = 0; e < E; ++e) prop_q[e] = T_pq[e];
                    prop_u = T_pu; prop_w = T_pw; prop_E = T_pE; prop_logp = T_plogp;
                }
                log_size = rfl(tn_logaddexp(log_size, T_ls));
#pragma unroll
                for (int e = 0; e < E; ++e) p_sum[e] += T_ps[e];  // :86 (in place: the aliases below see the new value)
                bool turn;
                {
                    // leftmost / rightmost halves: nuts.py:56-69 (the first leaf of the new subtree has momentum T_lp, the last is cur)
                    double T_lv[E], t1[E], t2[E];
                    vel(T_lp, T_lv);
                    double lm_begin_v[E], lm_end_p[E], lm_end_v[E], rm_begin_p[E], rm_begin_v[E], rm_end_v[E], lm_ps[E], rm_ps[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        lm_begin_v[e] = dir > 0 ? old_left.v[e] : cur.v[e];
                        lm_end_p[e] = dir > 0 ? old_right.p[e] : T_lp[e];
                        lm_end_v[e] = dir > 0 ? old_right.v[e] : T_lv[e];
                        rm_begin_p[e] = dir > 0 ? T_lp[e] : old_left.p[e];
                        rm_begin_v[e] = dir > 0 ? T_lv[e] : old_left.v[e];
                        rm_end_v[e] = dir > 0 ? cur.v[e] : old_right.v[e];
                        lm_ps[e] = dir > 0 ? p_sum[e] : T_ps[e];
                        rm_ps[e] = dir > 0 ? T_ps[e] : p_sum[e];
                    }
#pragma unroll
                    for (int e = 0; e < E; ++e) { t1[e] = lm_ps[e] + rm_begin_p[e]; t2[e] = lm_end_p[e] + rm_ps[e]; }
                    turn = dot6(p_sum, left.v, p_sum, right.v, t1, lm_begin_v, t1, rm_begin_v, t2, lm_end_v, t2, rm_end_v);
                }
                turning = turn ? 1 : 0;
                if (turning) break;
            }
            if (err) break;
            // ---- iteration end: base_hmc.py:252-262 ----
            const double accept_stat = accept_sum / (double)n_prop;
            if (warm && a.cfg.adapt_step_size) {  // step_size.py:31-45
                const double wgt = 1. / (count + a.cfg.t_0);
                hbar = ((1. - wgt) * hbar + wgt * (a.cfg.target_accept - accept_stat));
                log_step = smu - hbar * sqrt(count) / a.cfg.gamma;
                const double mk = exp(-a.cfg.k * log(count));
                log_bar = mk * log_step + (1. - mk) * log_bar;
                count += 1.;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) qc[e] = prop_q[e];
            u_cur = prop_u;
            const int orow = i_iter - a.iter_out0;
            if (orow >= 0 && orow < a.n_out) {
                if (lane == 0) {
                    double *st = a.stats + ((size_t)chain * a.n_out + orow) * BFHIP_STAT_STRIDE;
                    st[BFHIP_NS_LOGP] = prop_logp;
                    st[BFHIP_NS_ENERGY] = prop_E;
                    st[BFHIP_NS_TREE_DEPTH] = (double)depth;
                    st[BFHIP_NS_TREE_SIZE] = (double)n_prop;
                    st[BFHIP_NS_MEAN_TREE_ACCEPT] = accept_stat;
                    st[BFHIP_NS_STEP_SIZE] = exp(log_step);
                    st[BFHIP_NS_STEP_SIZE_BAR] = exp(log_bar);
                    st[BFHIP_NS_WARMUP] = warm ? 1. : 0.;
                    st[BFHIP_NS_ENERGY_CHANGE] = prop_E - start.energy;
                    st[BFHIP_NS_MAX_ENERGY_CHANGE] = max_de;
                    st[BFHIP_NS_DIVERGING] = (double)diverging;
                    double *tt = a.stats_t + ((size_t)chain * a.n_out + orow) * 2;
                    tt[0] = prop_u;
                    tt[1] = (prop_w == 0) ? 1. : prop_w / expm1(prop_w);
                }
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (in[e]) a.samples[((size_t)chain * a.n_out + orow) * d + lane * E + e] = qc[e];
            }
            if (warm && a.cfg.adapt_metric) {
                const long delta = (long)(n_samples - prev_upd);
                if constexpr (FULLM) {
                    // QuadMetricFullAdapt.update: metrics.py:294-324, _WeightedCovariance.add_sample :401-407 (as in bfhip_sampler.hip)
                    double fm[E], bm[E], od[E], nd[E];
                    load_vec(BFHIP_VEC_FG_MEAN, fm, 0.);
                    load_vec(BFHIP_VEC_BG_MEAN, bm, 0.);
                    double *fgT = matp + BF_MAT_FG * msz, *bgT = matp + BF_MAT_BG * msz, *covT = matp + BF_MAT_COV * msz;
                    fg_n += 1.;
#pragma unroll
                    for (int e = 0; e < E; ++e) { od[e] = qc[e] - fm[e]; fm[e] += od[e] / fg_n; nd[e] = qc[e] - fm[e]; }
                    const bool refresh = (delta + 1) % (long)a.cfg.update_window == 0;   // _update_from_weightvar: :287-292
                    bf_welford_cov<E>(fgT, nd, od, d, lane, refresh ? covT : nullptr, fg_n);
                    bg_n += 1.;
#pragma unroll
                    for (int e = 0; e < E; ++e) { od[e] = qc[e] - bm[e]; bm[e] += od[e] / bg_n; nd[e] = qc[e] - bm[e]; }
                    bf_welford_cov<E>(bgT, nd, od, d, lane);
                    if (refresh) {
                        double *wT = matp + BF_MAT_WORK * msz;
                        if (bf_chol_rows<E>(covT, wT, d, lane))
                            bf_chol_publish<E>(wT, matp + BF_MAT_CHOL * msz, matp + BF_MAT_CHOL_ROWS * msz, d, lane);
                    }
                    if ((double)delta >= adapt_window) {
                        for (int j = 0; j < d; ++j) {
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const int i = lane * E + e;
                                if (i < d) {
                                    fgT[(size_t)j * d + i] = bgT[(size_t)j * d + i];
                                    bgT[(size_t)j * d + i] = (i == j) ? 10. : 0.;  // _WeightedCovariance(n): 10 I
                                }
                            }
                        }
#pragma unroll
                        for (int e = 0; e < E; ++e) { fm[e] = bm[e]; bm[e] = 0.; }
                        fg_n = bg_n; bg_n = 10.; prev_upd = n_samples;
                        if (a.cfg.doubling) adapt_window *= 2.;
                    }
                    n_samples += 1.;
                    store_vec(BFHIP_VEC_FG_MEAN, fm);
                    store_vec(BFHIP_VEC_BG_MEAN, bm);
                } else {
                    // QuadMetricDiagAdapt.update: metrics.py:186-211
                    double fm[E], fr[E], bm[E], br[E];
                    load_vec(BFHIP_VEC_FG_MEAN, fm, 0.); load_vec(BFHIP_VEC_FG_RAW, fr, 0.);
                    load_vec(BFHIP_VEC_BG_MEAN, bm, 0.); load_vec(BFHIP_VEC_BG_RAW, br, 0.);
                    fg_n += 1.; bg_n += 1.;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        double od = qc[e] - fm[e]; fm[e] += od / fg_n; fr[e] += 1. * od * (qc[e] - fm[e]);
                        od = qc[e] - bm[e]; bm[e] += od / bg_n; br[e] += 1. * od * (qc[e] - bm[e]);
                    }
                    if ((delta + 1) % (long)a.cfg.update_window == 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            if (in[e]) var[e] = fr[e] / fg_n;
                        store_vec(BFHIP_VEC_VAR, var);
                    }
                    if ((double)delta >= adapt_window) {
#pragma unroll
                        for (int e = 0; e < E; ++e) { fm[e] = bm[e]; fr[e] = br[e]; bm[e] = 0.; br[e] = 0.; }
                        fg_n = bg_n; bg_n = 10.; prev_upd = n_samples;
                        if (a.cfg.doubling) adapt_window *= 2.;
                    }
                    n_samples += 1.;
                    store_vec(BFHIP_VEC_FG_MEAN, fm); store_vec(BFHIP_VEC_FG_RAW, fr);
                    store_vec(BFHIP_VEC_BG_MEAN, bm); store_vec(BFHIP_VEC_BG_RAW, br);
                }
            }
            i_iter += 1;
        }
        // ---- write the chain state back ----
        store_vec(BFHIP_VEC_Q, qc);
        if (lane == 0) {
            for (int k = 0; k < 4; ++k) a.rng[(size_t)chain * 4 + k] = rs[k];
            scp[BFHIP_SC_LOG_STEP] = log_step; scp[BFHIP_SC_LOG_BAR] = log_bar; scp[BFHIP_SC_HBAR] = hbar; scp[BFHIP_SC_COUNT] = count;
            scp[BFHIP_SC_FG_N] = fg_n; scp[BFHIP_SC_BG_N] = bg_n; scp[BFHIP_SC_N_SAMPLES] = n_samples;
            scp[BFHIP_SC_PREV_UPDATE] = prev_upd; scp[BFHIP_SC_ADAPT_WINDOW] = adapt_window;
            scp[BFHIP_SC_I_ITER] = (double)i_iter; scp[BFHIP_SC_ERROR] = (double)err;
            a.tu[chain] = u_cur;
            if (a.n_leapfrog && nlf) atomicAdd(a.n_leapfrog, nlf);
        }
    }
    // the chains of this wave's workgroup that are still running need its matvec jobs, its share of the contractions and the barriers
    {
        double z[E], t0[E], t1[E], t2[E], t3[E];
#pragma unroll
        for (int e = 0; e < E; ++e) z[e] = 0.;
        while (rdv_matvec(false, z, z, z, z, t0, t1, t2, t3)) {
            if (f_pld) rdv_pld();
        }
    }
}

template <int W, bool FULLM>
static int tg_launch_t(bfhip_ctx *ctx, const TnutsArgs &a, const double *mat) {
    const DevModel &m = ctx->model;
    const TgLds LL = tg_lds_layout(16 * W, m.pld);
    const size_t lds = LL.total * sizeof(double);
    if (lds > (size_t)160 * 1024)
        return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_tnuts_run: this pipeline density needs %zu KB of LDS in the tempered kernel (160 KB)", lds / 1024);
    auto k = bf_tnuts_gen_kernel<W, FULLM>;
    if (lds > 64 * 1024) BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((a.n_chain + a.cpg - 1) / a.cpg), dim3(64 * TG_WAVES), lds, ctx->stream, m, a, (double *)mat);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

int bf_tnuts_gen_launch(bfhip_ctx *ctx, const TnutsArgs &a_in, const double *mat) {
    const DevModel &m = ctx->model;
    TnutsArgs a = a_in;
    const int DP = m.DP;
    // the subtree stack: 4 TN_MAXL vector slots of DP doubles per chain
    const size_t need = (size_t)((a.n_chain + 15) / 16 * 16) * (4 * TN_MAXL) * DP * sizeof(double);
    if (ctx->scratch_bytes < need) {
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) BF_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = NULL;
        ctx->scratch_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    a.scratch = (double *)ctx->scratch;
    const bool full = mat != NULL;
#define TG_PICK(Wv) (full ? tg_launch_t<Wv, true>(ctx, a, mat) : tg_launch_t<Wv, false>(ctx, a, mat))
    switch (DP) {
    case 16: return TG_PICK(1);
    case 32: return TG_PICK(2);
    case 64: return TG_PICK(4);
    case 128: return TG_PICK(8);
    }
#undef TG_PICK
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_tnuts_run: padded dimension %d", DP);
}
