// bfhip_sampler.hip -- fused NUTS / HMC transitions for many chains (gfx950).
//
// Work decomposition ("tile phase / chain phase"):
//   * A workgroup of W = DP/16 wavefronts owns a GROUP of 16 chains for the whole launch.
//   * Tile phase (gradient): the batched matvecs G^T = S X^T and (H (X-mu)^T) for the 16 chains run on
//     v_mfma_f64_16x16x4_f64.  Wave w computes output rows 16w..16w+15 for all 16 chains; its A operands
//     (one 16 x DP row tile of S and of H) stay in registers for the whole launch.
//   * Chain phase (everything O(d)): wave w owns chains w*CPW .. w*CPW+CPW-1 (CPW = 16/W); a chain is a
//     ROW of RW = 4W consecutive lanes, lane j of the row holds dimensions 4j..4j+3 of every state vector.
//     Dot products are row reductions (no LDS, no barrier); per-chain scalars are replicated over the row.
//   * The two layouts meet in LDS: XB (B operands, written by the chain phase) and GB (matvec results).
//     Two workgroup barriers per trip, none inside the tree logic.
//
// Every chain is an independent state machine (INIT -> LEAF ... -> iteration end -> INIT ...); one loop
// trip evaluates ONE gradient for all 16 chains of the group, whatever each chain needs it for.  The
// compute_state() call that opens every iteration (base_hmc.py:70) is a leapfrog with epsilon = 0, so
// chains never wait for each other inside an iteration or across iterations.
//
// The recursion of Tree._build_subtree (samplers/nuts.py:134-178) is flattened: leaf i of a 2^depth
// subtree is merged upwards while bit `level` of i is set; completed sub-subtrees wait on a per-chain
// stack (vectors in global scratch, scalars in LDS).  Random draws are consumed in the recursion's
// post-order, so a chain reproduces the CPU oracle's trajectory for the same xoshiro stream.
#include "bfhip_eval.h"

struct SamplerArgs {
    bfhip_sampler_config cfg;
    int n_chain, iter_end, iter_out0, n_out, nslot;
    uint64_t *rng;
    double *sc, *vec, *samples, *stats;
    unsigned long long *n_leapfrog;
    double *scratch;
    double *dbg;      // optional trace of one chain: [dbg_cap][32] doubles (diagnostics only)
    int dbg_chain, dbg_cap;
};

enum { M_INIT = 0, M_LEAF = 1, M_OOB = 2, M_DONE = 3 };
enum { SL_LEFT_Q = 0, SL_LEFT_P, SL_LEFT_G, SL_RIGHT_Q, SL_RIGHT_P, SL_RIGHT_G, SL_PROP_Q, SL_PSUM, SL_STACK };
enum { LS_LS = 0, LS_E, LS_LOGP, LS_ACC, LS_N };

__device__ inline void ld4(const double *p, double (&v)[4]) {
    const d2_t a = ((const d2_t *)p)[0], b = ((const d2_t *)p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ inline void st4(double *p, const double (&v)[4]) {
    d2_t a, b;
    a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
    ((d2_t *)p)[0] = a;
    ((d2_t *)p)[1] = b;
}

template <int RW>
__device__ inline double row_sum(double v) {
#pragma unroll
    for (int msk = RW / 2; msk >= 1; msk >>= 1) v += __shfl_xor(v, msk, 64);
    return v;
}

template <int W, bool NUTS>
__global__ __launch_bounds__(64 * W) void bf_sampler_kernel(DevModel m, SamplerArgs a) {
    constexpr int DP = 16 * W, NS = 4 * W, RW = 4 * W, CPW = 16 / W;
    constexpr int XS = 65;       // XB row stride (doubles): odd => conflict-free column writes
    constexpr int GS = DP + 1;   // GB row stride
    constexpr int MAXL = BFHIP_MAX_TREEDEPTH;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *XB = lds;                        // [3][NS][XS]
    double *GB = XB + 3 * NS * XS;           // [3][16][GS]
    double *PDL = GB + 3 * 16 * GS;          // [PD_N][DP]
    double *LS = PDL + PD_N * DP;            // [MAXL][LS_N][16]
    int *alive = (int *)(LS + MAXL * LS_N * 16);  // [2]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // tile-phase identity
    const int mc = lane & 15, mg = lane >> 4;
    // chain-phase identity
    const int row = lane / RW, j = lane % RW;
    const int cl = w * CPW + row;
    const int chain = blockIdx.x * 16 + cl;
    const bool real = chain < a.n_chain;
    const int d = m.d;

    for (int i = tid; i < PD_N * DP; i += 64 * W) PDL[i] = m.pd[i];
    if (tid < 2) alive[tid] = 0;

    // A operands of this wave's row tile, resident in registers for the whole launch
    double Sreg[NS], Hreg[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        Sreg[s] = m.has_quad ? m.Sf[(w * NS + s) * 64 + lane] : 0.;
        Hreg[s] = m.use_bound ? m.Hf[(w * NS + s) * 64 + lane] : 0.;
    }

    // ---- per-chain state ----
    double q[4], p[4] = {0., 0., 0., 0.}, g[4] = {0., 0., 0., 0.}, var[4];
    double TLp[4], TPs[4], TPq[4];
    uint64_t rs[4] = {0, 0, 0, 0};
    double log_step = 0., log_bar = 0., hbar = 0., smu = 0., count = 1.;
    int i_iter = 0, mode = M_DONE, prev_mode = M_INIT, err = 0;
    double eps = 0., eps_t = 0., beta_saved = 0.;
    int dir = 1, depth = 0, i_leaf = 0, n_prop = 0, diverged = 0;
    double start_energy = 0., tree_ls = 0., acc_sum = 0., max_dE = 0., prop_E = 0., prop_logp = 0.;
    double T_ls = 0., T_E = 0., T_logp = 0., T_acc = 0.;
    unsigned long long nlf = 0;
    double *sbase = a.scratch + ((size_t)(real ? chain : 0) * a.nslot) * DP + 4 * j;
    double *scp = a.sc + (size_t)(real ? chain : 0) * BFHIP_SC_N;
    double *vecp = a.vec + (size_t)(real ? chain : 0) * BFHIP_VEC_N * d;
    const int nw = a.cfg.n_warmup;

    auto load_vec = [&](int field, double (&v)[4], double pad) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int dim = 4 * j + e;
            v[e] = (dim < d) ? vecp[field * d + dim] : pad;
        }
    };
    auto store_vec = [&](int field, const double (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int dim = 4 * j + e;
            if (dim < d) vecp[field * d + dim] = v[e];
        }
    };
    // metric.random: samplers/hmc_utils/metrics.py:83-86.  One xoshiro draw K keys a SplitMix64 counter
    // stream; pair P of the stream gives dimensions 2P (cos) and 2P+1 (sin) by Box-Muller.
    auto draw_momentum = [&]() {
        const uint64_t K = bf_xoshiro_next(rs);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint64_t P = (uint64_t)(2 * j + h);
            const double u1 = bf_u01_open0(bf_mix64(K + (2 * P + 1) * BF_GOLDEN));
            const double u2 = bf_u01(bf_mix64(K + (2 * P + 2) * BF_GOLDEN));
            const double rad = sqrt(-2. * log(u1));
            const double th = BF_TWO_PI * u2;
            const double z0 = rad * cos(th), z1 = rad * sin(th);
            const int d0 = 4 * j + 2 * h;
            p[2 * h] = (d0 < d) ? (1. / sqrt(var[2 * h])) * z0 : 0.;
            p[2 * h + 1] = (d0 + 1 < d) ? (1. / sqrt(var[2 * h + 1])) * z1 : 0.;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = 0.;
    };

    if (real) {
        for (int k = 0; k < 4; ++k) rs[k] = a.rng[(size_t)chain * 4 + k];
        log_step = scp[BFHIP_SC_LOG_STEP];
        log_bar = scp[BFHIP_SC_LOG_BAR];
        hbar = scp[BFHIP_SC_HBAR];
        smu = scp[BFHIP_SC_MU];
        count = scp[BFHIP_SC_COUNT];
        i_iter = (int)scp[BFHIP_SC_I_ITER];
        err = (int)scp[BFHIP_SC_ERROR];
        load_vec(BFHIP_VEC_Q, q, 0.);
        load_vec(BFHIP_VEC_VAR, var, 1.);
        if (i_iter < a.iter_end && err == 0) {
            mode = M_INIT;
            draw_momentum();
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { q[e] = 0.; p[e] = 0.; g[e] = 0.; var[e] = 1.; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { TLp[e] = 0.; TPs[e] = 0.; TPq[e] = 0.; }
    __syncthreads();

    // uniform draw of this chain's stream; logbern(l) = log(U) < l (samplers/nuts.py:200-203)
    auto logbern = [&](double l) -> bool {
        if (l != l) err = 2;
        return log(bf_u01(bf_xoshiro_next(rs))) < l;
    };

    for (int trip = 0;; ++trip) {
        // ================= phase A: first half of the leapfrog, B operands =================
        double xs[4], jac[4], gj[4], xo[4];
        double logdet = 0.;
        const bool evaluating = mode != M_DONE;
        if (evaluating) {
            if (mode != M_OOB) {
                eps_t = (mode == M_LEAF) ? eps * (double)dir : 0.;
                const double dt = 0.5 * eps_t;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = p[e] + dt * g[e];               // integration.py:80
                    q[e] = q[e] + eps_t * (var[e] * p[e]); // :82-85
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int dim = 4 * j + e;
                xo[e] = q[e];
                jac[e] = 1.;
                gj[e] = 0.;
                if (m.has_transform) {
                    double J, J2;
                    bf_to_original(q[e], (int)PDL[PD_KIND * DP + dim], PDL[PD_LO * DP + dim], PDL[PD_RG * DP + dim],
                                   xo[e], J, J2);
                    logdet += log(fabs(J));
                    jac[e] = J;
                    gj[e] = J2 / J;
                }
                xs[e] = m.has_su ? (xo[e] - PDL[PD_SU_LO * DP + dim]) / PDL[PD_SU_DIFF * DP + dim] : xo[e];
                const double mu = PDL[PD_MU * DP + dim];
                double x_eval = xs[e];
                if (mode == M_OOB)  // modules/poly.py:482
                    x_eval = (m.alpha * xs[e] + (beta_saved - m.alpha) * mu) / beta_saved;
                XB[(0 * NS + j) * XS + cl + 16 * e] = x_eval;
                if (m.use_bound) XB[(1 * NS + j) * XS + cl + 16 * e] = xs[e] - mu;
                if (m.use_decay) XB[(2 * NS + j) * XS + cl + 16 * e] = xo[e] - PDL[PD_DMU * DP + dim];
            }
            alive[trip & 1] = 1;
        }
        __syncthreads();  // B1
        if (alive[trip & 1] == 0) break;  // every chain of the group is done (uniform)
        if (tid == 0) alive[(trip + 1) & 1] = 0;

        // ================= phase B: gradient tile on MFMA =================
        {
            d4_t accS = {0., 0., 0., 0.}, accH = {0., 0., 0., 0.}, accD = {0., 0., 0., 0.};
            if (m.has_quad) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    accS = __builtin_amdgcn_mfma_f64_16x16x4f64(Sreg[s], XB[(0 * NS + s) * XS + lane], accS, 0, 0, 0);
            }
            if (m.use_bound) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    accH = __builtin_amdgcn_mfma_f64_16x16x4f64(Hreg[s], XB[(1 * NS + s) * XS + lane], accH, 0, 0, 0);
            }
            if (m.use_decay) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    accD = __builtin_amdgcn_mfma_f64_16x16x4f64(m.Hdf[(w * NS + s) * 64 + lane],
                                                                XB[(2 * NS + s) * XS + lane], accD, 0, 0, 0);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int dim = 16 * w + 4 * r4 + mg;
                GB[(0 * 16 + mc) * GS + dim] = accS[r4];
                if (m.use_bound) GB[(1 * 16 + mc) * GS + dim] = accH[r4];
                if (m.use_decay) GB[(2 * 16 + mc) * GS + dim] = accD[r4];
            }
        }
        __syncthreads();  // B2

        // ================= phase C: finish the evaluation =================
        double gn[4], hv[4], dgr[4], xm[4];
        double logp_new = 0., E_new = 0.;
        bool have_eval = false;
        if (evaluating) {
            double red[5] = {0., 0., 0., 0., 0.};  // quad (quad_0), lin (lin_0), beta^2, dot(jj_0, x-mu), decay beta^2
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int dim = 4 * j + e;
                const double c = PDL[PD_LIN * DP + dim];
                const double mu = PDL[PD_MU * DP + dim];
                double sx = GB[(0 * 16 + cl) * GS + dim];
                hv[e] = m.use_bound ? GB[(1 * 16 + cl) * GS + dim] : 0.;
                dgr[e] = m.use_decay ? GB[(2 * 16 + cl) * GS + dim] : 0.;
                xm[e] = xs[e] - mu;
                double x_eval = xs[e];
                if (mode == M_OOB) x_eval = (m.alpha * xs[e] + (beta_saved - m.alpha) * mu) / beta_saved;
                red[0] += x_eval * sx;
                red[1] += c * x_eval;
                gn[e] = sx + c;
                red[2] += xm[e] * hv[e];
                if (mode == M_OOB) red[3] += gn[e] * xm[e];  // dot(jj_0, x - mu), poly.py:496
                red[4] += (xo[e] - PDL[PD_DMU * DP + dim]) * dgr[e];
            }
            red[0] = row_sum<RW>(red[0]);
            red[1] = row_sum<RW>(red[1]);
            if (m.use_bound) red[2] = row_sum<RW>(red[2]);
            if (m.use_bound) red[3] = row_sum<RW>(red[3]);
            if (m.use_decay) red[4] = row_sum<RW>(red[4]);
            if (m.has_transform) logdet = row_sum<RW>(logdet);

            double f = (m.c0 + red[1]) + 0.5 * red[0];
            const double beta = sqrt(red[2]);
            bool oob_now = false;
            if (m.use_bound) {
                if (mode == M_OOB) {  // second pass: f, gn currently hold f_0 and jj_0 (poly.py:484-496)
                    const double f0 = f;
                    f = (beta_saved * f0 - (beta_saved - m.alpha) * m.f_mu) / m.alpha;
                    const double coef = (f0 - m.f_mu) / m.alpha - red[3] / beta_saved;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gn[e] = gn[e] + coef * (hv[e] / beta_saved);
                } else if (beta > m.alpha) {
                    oob_now = true;
                }
            }
            if (oob_now) {
                // outside the alpha-ellipsoid: spend one more trip on the projected point x_0
                beta_saved = beta;
                prev_mode = mode;
                mode = M_OOB;
            } else {
                if (mode == M_OOB) mode = prev_mode;
                // chain rule (module.py:226, density.py:558), decay (:740-746), transform (:747-750)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (m.has_su) gn[e] = gn[e] / PDL[PD_SU_DIFF * DP + 4 * j + e];
                    gn[e] = gn[e] * jac[e];
                }
                if (m.use_decay) {
                    f -= m.decay_gamma * bf_clip0(red[4] - m.decay_alpha2);
                    if (red[4] > m.decay_alpha2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) gn[e] -= 2. * m.decay_gamma * dgr[e];
                    }
                }
                if (m.has_transform) {
                    f += logdet;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gn[e] += gj[e];
                }
                logp_new = f;
                have_eval = true;
            }
        }
        // second half of the leapfrog and the kinetic energy
        {
            const bool need_kin = have_eval;
            double kin = 0.;
            const double dt = 0.5 * eps_t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double pe = p[e] + dt * gn[e];  // integration.py:90
                kin += pe * (var[e] * pe);            // metrics.py:88-91
                if (need_kin) p[e] = pe;
            }
            if (__any(need_kin)) kin = row_sum<RW>(kin);
            E_new = 0.5 * kin - logp_new;            // integration.py:92-93
        }

        // ================= per-chain state machine =================
        double dbgv[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
        bool it_end = false;   // this chain finished an iteration in this trip
        bool complete = false; // NUTS: subtree of this doubling is complete
        const bool is_leaf = have_eval && mode == M_LEAF;  // (an INIT trip becomes M_LEAF below; it is not a leaf)
        if (have_eval) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = gn[e];
            if (mode == M_INIT) {
                // BaseHMC.astep start: base_hmc.py:70-76, Tree.__init__: nuts.py:24-43
                if (!(fabs(E_new) <= 1.7976931348623157e308)) {
                    err = 1;
                    mode = M_DONE;
                } else {
                    start_energy = E_new;
                    st4(sbase + SL_LEFT_Q * DP, q);
                    st4(sbase + SL_LEFT_P * DP, p);
                    st4(sbase + SL_LEFT_G * DP, g);
                    st4(sbase + SL_RIGHT_Q * DP, q);
                    st4(sbase + SL_RIGHT_P * DP, p);
                    st4(sbase + SL_RIGHT_G * DP, g);
                    st4(sbase + SL_PROP_Q * DP, q);
                    st4(sbase + SL_PSUM * DP, p);
                    prop_E = E_new;
                    prop_logp = logp_new;
                    depth = 0;
                    tree_ls = 0.;
                    acc_sum = 0.;
                    n_prop = 0;
                    max_dE = 0.;
                    diverged = 0;
                    i_leaf = 0;
                    eps = exp(i_iter < nw ? log_step : log_bar);  // step_size.py:25-29
                    dir = 1;
                    if (NUTS) dir = logbern(-0.6931471805599453094) ? 1 : -1;  // nuts.py:210
                    mode = M_LEAF;
                }
            } else if (mode == M_LEAF) {
                nlf += 1;
                if (NUTS) {
                    // ---- Tree._single_step: nuts.py:105-132 ----
                    n_prop += 1;
                    double dE = E_new - start_energy;
                    if (dE != dE) dE = INFINITY;
                    if (fabs(dE) > fabs(max_dE)) max_dE = dE;
                    if (fabs(dE) < a.cfg.max_change) {
                        const double pacc = exp(-dE);
                        T_acc = pacc > 1. ? 1. : pacc;
                        T_ls = -dE;
                        T_E = E_new;
                        T_logp = logp_new;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { TLp[e] = p[e]; TPs[e] = p[e]; TPq[e] = q[e]; }
                    } else {
                        diverged = 1;
                        T_acc = 0.;
                    }
                } else {
                    i_leaf += 1;
                }
            }
        }

        if (NUTS) {
            // ---- Tree._build_subtree merges (nuts.py:134-178), iteratively ----
            const bool leaf_ok = is_leaf && !diverged;
            int lev = 0;
            bool turned = false;
            while (true) {
                const bool do_m = leaf_ok && !turned && lev < depth && ((i_leaf >> lev) & 1);
                if (!__any(do_m)) break;
                double A[4], B[4], S1[4], dts[6] = {0., 0., 0., 0., 0., 0.};
                double *slot = sbase + (SL_STACK + 4 * lev) * DP;
                if (do_m) {
                    ld4(slot + 0 * DP, A);
                    ld4(slot + 1 * DP, B);
                    ld4(slot + 2 * DP, S1);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { A[e] = 0.; B[e] = 0.; S1[e] = 0.; }
                }
                double psum[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    psum[e] = S1[e] + TPs[e];
                    const double vA = var[e] * A[e], vB = var[e] * B[e], vC = var[e] * TLp[e], vD = var[e] * p[e];
                    dts[0] += psum[e] * vA;  // nuts.py:150-151
                    dts[1] += psum[e] * vD;
                    const double ps1 = S1[e] + TLp[e];  // :155-157
                    dts[2] += ps1 * vA;
                    dts[3] += ps1 * vC;
                    const double ps2 = B[e] + TPs[e];   // :158-160
                    dts[4] += ps2 * vB;
                    dts[5] += ps2 * vD;
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) dts[k] = row_sum<RW>(dts[k]);
                if (do_m) {
                    bool turning = (dts[0] <= 0.) || (dts[1] <= 0.);
                    if (lev >= 1) turning = turning || (dts[2] <= 0.) || (dts[3] <= 0.) || (dts[4] <= 0.) || (dts[5] <= 0.);
                    const double *lsp = LS + (lev * LS_N) * 16 + cl;
                    for (int k = 0; k < 6; ++k) dbgv[k] = dts[k];
                    dbgv[6] = lev; dbgv[7] = turning;
                    T_acc = lsp[LS_ACC * 16] + T_acc;  // :173
                    // nuts.py:163-167 run even when THIS merge's check says turning: the draw is consumed
                    const double ls1 = lsp[LS_LS * 16];
                    const double ls = bf_logaddexp(ls1, T_ls);
                    const bool keep_t2 = logbern(T_ls - ls);
                    if (turning) {
                        turned = true;
                    } else {
                        if (!keep_t2) {
                            ld4(slot + 3 * DP, TPq);
                            T_E = lsp[LS_E * 16];
                            T_logp = lsp[LS_LOGP * 16];
                        }
                        T_ls = ls;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { TLp[e] = A[e]; TPs[e] = psum[e]; }
                        lev += 1;
                    }
                }
            }
            if (is_leaf) {
                if (diverged || turned) {
                    // unwind: every pending ancestor adds its left half's accept_sum (nuts.py:173)
                    for (int al = (diverged ? 0 : lev + 1); al < depth; ++al)
                        if ((i_leaf >> al) & 1) T_acc = LS[(al * LS_N + LS_ACC) * 16 + cl] + T_acc;
                    depth += 1;          // nuts.py:71-73
                    acc_sum += T_acc;
                    it_end = true;
                } else if (lev == depth) {
                    complete = true;
                } else {
                    double *slot = sbase + (SL_STACK + 4 * lev) * DP;
                    st4(slot + 0 * DP, TLp);
                    st4(slot + 1 * DP, p);
                    st4(slot + 2 * DP, TPs);
                    st4(slot + 3 * DP, TPq);
                    double *lsp = LS + (lev * LS_N) * 16 + cl;
                    lsp[LS_LS * 16] = T_ls;
                    lsp[LS_E * 16] = T_E;
                    lsp[LS_LOGP * 16] = T_logp;
                    lsp[LS_ACC * 16] = T_acc;
                    i_leaf += 1;
                }
            }
            // ---- Tree.extend after a complete subtree: nuts.py:71-103 ----
            if (__any(complete)) {
                double oldL[4], oldR[4], ps[4], dts[6] = {0., 0., 0., 0., 0., 0.};
                bool swap = false;
                if (complete) {
                    depth += 1;
                    acc_sum += T_acc;
                    swap = logbern(T_ls - tree_ls);  // :81-83
                    tree_ls = bf_logaddexp(tree_ls, T_ls);  // :85
                    ld4(sbase + SL_PSUM * DP, ps);
                    ld4(sbase + SL_LEFT_P * DP, oldL);
                    ld4(sbase + SL_RIGHT_P * DP, oldR);
                    if (swap) {
                        st4(sbase + SL_PROP_Q * DP, TPq);
                        prop_E = T_E;
                        prop_logp = T_logp;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { oldL[e] = 0.; oldR[e] = 0.; ps[e] = 0.; }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ps[e] += TPs[e];  // :86 (in place)
                    const double vN = var[e] * p[e], vT = var[e] * TLp[e], vL = var[e] * oldL[e], vR = var[e] * oldR[e];
                    // NOTE (reference behaviour, kept on purpose): leftmost_p_sum (dir > 0) / rightmost_p_sum
                    // (dir < 0) alias self.p_sum, which line 86 has just updated in place.
                    if (dir > 0) {
                        dts[0] += ps[e] * vL;  // left = old left
                        dts[1] += ps[e] * vN;  // right = new end
                        const double ps1 = ps[e] + TLp[e];    // (aliased) leftmost_p_sum + rightmost_begin.p
                        dts[2] += ps1 * vL;                   // leftmost_begin = old left
                        dts[3] += ps1 * vT;                   // rightmost_begin = tree.left
                        const double ps2 = oldR[e] + TPs[e];  // leftmost_end.p + rightmost_p_sum
                        dts[4] += ps2 * vR;                   // leftmost_end = old right
                        dts[5] += ps2 * vN;                   // rightmost_end = tree.right
                    } else {
                        dts[0] += ps[e] * vN;  // left = new end
                        dts[1] += ps[e] * vR;  // right = old right
                        const double ps1 = TPs[e] + oldL[e];  // leftmost_p_sum + rightmost_begin.p
                        dts[2] += ps1 * vN;                   // leftmost_begin = tree.right
                        dts[3] += ps1 * vL;                   // rightmost_begin = old left
                        const double ps2 = TLp[e] + ps[e];    // leftmost_end.p + (aliased) rightmost_p_sum
                        dts[4] += ps2 * vT;                   // leftmost_end = tree.left
                        dts[5] += ps2 * vR;                   // rightmost_end = old right
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) dts[k] = row_sum<RW>(dts[k]);
                if (complete) {
                    st4(sbase + SL_PSUM * DP, ps);
                    const int eo = (dir > 0) ? SL_RIGHT_Q : SL_LEFT_Q;
                    st4(sbase + (eo + 0) * DP, q);
                    st4(sbase + (eo + 1) * DP, p);
                    st4(sbase + (eo + 2) * DP, g);
                    bool turning = false;
#pragma unroll
                    for (int k = 0; k < 6; ++k) turning = turning || (dts[k] <= 0.);
                    if (turning || depth >= a.cfg.max_treedepth) {
                        it_end = true;
                    } else {
                        const int nd = logbern(-0.6931471805599453094) ? 1 : -1;  // nuts.py:210
                        if (nd != dir) {
                            const int eo2 = (nd > 0) ? SL_RIGHT_Q : SL_LEFT_Q;
                            ld4(sbase + (eo2 + 0) * DP, q);
                            ld4(sbase + (eo2 + 1) * DP, p);
                            ld4(sbase + (eo2 + 2) * DP, g);
                        }
                        dir = nd;
                        i_leaf = 0;
                    }
                }
            }
        }

        // ---- HMC._hamiltonian_step end of trajectory: samplers/hmc.py:21-49 ----
        double h_accept_stat = 0., h_dE = 0.;
        int h_accepted = 0;
        if (!NUTS) {
            if (is_leaf && i_leaf >= a.cfg.n_int_step) {
                const bool fin = fabs(E_new) <= 1.7976931348623157e308;
                h_dE = fin ? (start_energy - E_new) : -INFINITY;
                diverged = (!fin || fabs(h_dE) > a.cfg.max_change) ? 1 : 0;
                h_accept_stat = exp(h_dE);
                if (h_accept_stat > 1.) h_accept_stat = 1.;
                if (!diverged) h_accepted = !(bf_u01(bf_xoshiro_next(rs)) >= h_accept_stat);
                if (h_accepted) st4(sbase + SL_PROP_Q * DP, q);
                prop_E = E_new;
                prop_logp = logp_new;
                it_end = true;
            }
        }

        // ================= iteration end: base_hmc.py:80-85 =================
        if (__any(it_end)) {
            if (it_end) {
                const bool warm = i_iter < nw;
                const double accept_stat = NUTS ? acc_sum / (double)n_prop : h_accept_stat;  // nuts.py:186
                if (warm && a.cfg.adapt_step_size) {  // step_size.py:31-45
                    const double wgt = 1. / (count + a.cfg.t_0);
                    hbar = ((1. - wgt) * hbar + wgt * (a.cfg.target_accept - accept_stat));
                    log_step = smu - hbar * sqrt(count) / a.cfg.gamma;
                    const double mk = pow(count, -a.cfg.k);
                    log_bar = mk * log_step + (1. - mk) * log_bar;
                    count += 1.;
                }
                const int orow = i_iter - a.iter_out0;
                const bool wr = orow >= 0 && orow < a.n_out;
                if (wr && j == 0) {
                    double *st = a.stats + ((size_t)chain * a.n_out + orow) * BFHIP_STAT_STRIDE;
                    if (NUTS) {
                        st[BFHIP_NS_LOGP] = prop_logp;
                        st[BFHIP_NS_ENERGY] = prop_E;
                        st[BFHIP_NS_TREE_DEPTH] = (double)depth;
                        st[BFHIP_NS_TREE_SIZE] = (double)n_prop;
                        st[BFHIP_NS_MEAN_TREE_ACCEPT] = accept_stat;
                        st[BFHIP_NS_STEP_SIZE] = exp(log_step);
                        st[BFHIP_NS_STEP_SIZE_BAR] = exp(log_bar);
                        st[BFHIP_NS_WARMUP] = warm ? 1. : 0.;
                        st[BFHIP_NS_ENERGY_CHANGE] = prop_E - start_energy;
                        st[BFHIP_NS_MAX_ENERGY_CHANGE] = max_dE;
                        st[BFHIP_NS_DIVERGING] = (double)diverged;
                    } else {
                        st[BFHIP_HS_LOGP] = prop_logp;
                        st[BFHIP_HS_ENERGY] = prop_E;
                        st[BFHIP_HS_N_INT_STEP] = (double)a.cfg.n_int_step;
                        st[BFHIP_HS_ACCEPT_STAT] = accept_stat;
                        st[BFHIP_HS_ACCEPTED] = (double)h_accepted;
                        st[BFHIP_HS_STEP_SIZE] = exp(log_step);
                        st[BFHIP_HS_STEP_SIZE_BAR] = exp(log_bar);
                        st[BFHIP_HS_WARMUP] = warm ? 1. : 0.;
                        st[BFHIP_HS_ENERGY_CHANGE] = h_dE;
                        st[BFHIP_HS_DIVERGING] = (double)diverged;
                        st[10] = 0.;
                    }
                }
                // the new sample
                ld4(sbase + SL_PROP_Q * DP, q);
                if (wr) {
                    double *sp = a.samples + ((size_t)chain * a.n_out + orow) * d;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * j + e < d) sp[4 * j + e] = q[e];
                }
                // QuadMetricDiagAdapt.update: metrics.py:186-211, _WeightedVariance.add_sample :354-360
                if (warm && a.cfg.adapt_metric) {
                    double fg_n = scp[BFHIP_SC_FG_N], bg_n = scp[BFHIP_SC_BG_N];
                    double n_samples = scp[BFHIP_SC_N_SAMPLES], prev_upd = scp[BFHIP_SC_PREV_UPDATE];
                    double adapt_window = scp[BFHIP_SC_ADAPT_WINDOW];
                    const long delta = (long)(n_samples - prev_upd);
                    double fm[4], fr[4], bm[4], br[4];
                    load_vec(BFHIP_VEC_FG_MEAN, fm, 0.);
                    load_vec(BFHIP_VEC_FG_RAW, fr, 0.);
                    load_vec(BFHIP_VEC_BG_MEAN, bm, 0.);
                    load_vec(BFHIP_VEC_BG_RAW, br, 0.);
                    fg_n += 1.;
                    bg_n += 1.;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        double od = q[e] - fm[e];
                        fm[e] += od / fg_n;
                        fr[e] += 1. * od * (q[e] - fm[e]);
                        od = q[e] - bm[e];
                        bm[e] += od / bg_n;
                        br[e] += 1. * od * (q[e] - bm[e]);
                    }
                    if ((delta + 1) % (long)a.cfg.update_window == 0) {  // metrics.py:181-184
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (4 * j + e < d) var[e] = fr[e] / fg_n;
                        store_vec(BFHIP_VEC_VAR, var);
                    }
                    if ((double)delta >= adapt_window) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { fm[e] = bm[e]; fr[e] = br[e]; bm[e] = 0.; br[e] = 0.; }
                        fg_n = bg_n;
                        bg_n = 10.;
                        prev_upd = n_samples;
                        if (a.cfg.doubling) adapt_window *= 2.;
                    }
                    n_samples += 1.;
                    store_vec(BFHIP_VEC_FG_MEAN, fm);
                    store_vec(BFHIP_VEC_FG_RAW, fr);
                    store_vec(BFHIP_VEC_BG_MEAN, bm);
                    store_vec(BFHIP_VEC_BG_RAW, br);
                    if (j == 0) {
                        scp[BFHIP_SC_FG_N] = fg_n;
                        scp[BFHIP_SC_BG_N] = bg_n;
                        scp[BFHIP_SC_N_SAMPLES] = n_samples;
                        scp[BFHIP_SC_PREV_UPDATE] = prev_upd;
                        scp[BFHIP_SC_ADAPT_WINDOW] = adapt_window;
                    }
                }
                i_iter += 1;
                if (i_iter < a.iter_end && err == 0) {
                    mode = M_INIT;
                    draw_momentum();
                } else {
                    mode = M_DONE;
                }
            }
        }
        if (err != 0 && mode != M_DONE) mode = M_DONE;
        if (a.dbg && real && chain == a.dbg_chain && j == 0 && trip < a.dbg_cap) {
            double *t = a.dbg + (size_t)trip * 32;
            t[0] = have_eval; t[1] = is_leaf; t[2] = i_leaf; t[3] = depth; t[4] = dir; t[5] = E_new; t[6] = logp_new;
            t[7] = it_end; t[8] = complete; t[9] = T_acc; t[10] = acc_sum; t[11] = tree_ls; t[12] = T_ls; t[13] = i_iter;
            t[14] = q[0]; t[15] = p[0];
            for (int k = 0; k < 8; ++k) t[16 + k] = dbgv[k];
        }
    }

    // ---- write the chain state back ----
    if (real) {
        store_vec(BFHIP_VEC_Q, q);
        if (j == 0) {
            for (int k = 0; k < 4; ++k) a.rng[(size_t)chain * 4 + k] = rs[k];
            scp[BFHIP_SC_LOG_STEP] = log_step;
            scp[BFHIP_SC_LOG_BAR] = log_bar;
            scp[BFHIP_SC_HBAR] = hbar;
            scp[BFHIP_SC_COUNT] = count;
            scp[BFHIP_SC_I_ITER] = (double)i_iter;
            scp[BFHIP_SC_ERROR] = (double)err;
            if (a.n_leapfrog && nlf) atomicAdd(a.n_leapfrog, nlf);
        }
    }
}

static size_t sampler_lds_bytes(int W) {
    const int DP = 16 * W, NS = 4 * W;
    size_t dbl = (size_t)3 * NS * 65 + (size_t)3 * 16 * (DP + 1) + (size_t)PD_N * DP + (size_t)BFHIP_MAX_TREEDEPTH * LS_N * 16;
    return dbl * sizeof(double) + 16;
}

template <int W, bool NUTS>
static int launch_sampler(bfhip_ctx *ctx, const SamplerArgs &args) {
    auto k = bf_sampler_kernel<W, NUTS>;
    const size_t lds = sampler_lds_bytes(W);
    if (lds > 64 * 1024)
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int groups = (args.n_chain + 15) / 16;
    hipLaunchKernelGGL(k, dim3(groups), dim3(64 * W), lds, ctx->stream, ctx->model, args);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

static double *g_dbg_buf = NULL;
static int g_dbg_chain = 0, g_dbg_cap = 0;
// diagnostics hook (not part of include/bfhip.h): trace one chain's trips into a device buffer
extern "C" void bfhip_debug_trace(double *buf, int chain, int cap) { g_dbg_buf = buf; g_dbg_chain = chain; g_dbg_cap = cap; }

extern "C" int bfhip_sampler_run(bfhip_ctx *ctx, const bfhip_sampler_config *cfg, int n_chain, int iter_end,
                                 uint64_t *rng, double *sc, double *vec, int iter_out0, int n_out, double *samples,
                                 double *stats, unsigned long long *n_leapfrog) {
    if (!ctx || !cfg || n_chain < 0) return bf_set_error(BFHIP_ERR_ARG, "bfhip_sampler_run: invalid argument");
    if (!ctx->has_model) return bf_set_error(BFHIP_ERR_STATE, "bfhip_sampler_run: no density uploaded");
    if (n_chain == 0) return 0;
    if (!rng || !sc || !vec || (n_out > 0 && (!samples || !stats)) || n_out < 0)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_sampler_run: NULL state or output array");
    if (cfg->sampler != 0 && cfg->sampler != 1) return bf_set_error(BFHIP_ERR_ARG, "sampler should be 0 (NUTS) or 1 (HMC)");
    if (cfg->sampler == 0 && (cfg->max_treedepth < 1 || cfg->max_treedepth > BFHIP_MAX_TREEDEPTH))
        return bf_set_error(BFHIP_ERR_ARG, "max_treedepth should be in [1, %d]", BFHIP_MAX_TREEDEPTH);
    if (cfg->sampler == 1 && cfg->n_int_step < 1) return bf_set_error(BFHIP_ERR_ARG, "n_int_step should be a positive int");
    if (!(cfg->max_change > 0.) || cfg->update_window < 1 || cfg->n_warmup < 0)
        return bf_set_error(BFHIP_ERR_ARG, "invalid sampler configuration");
    const DevModel &m = ctx->model;
    const int W = m.DP / 16;
    SamplerArgs args;
    args.cfg = *cfg;
    args.n_chain = n_chain;
    args.iter_end = iter_end;
    args.iter_out0 = iter_out0;
    args.n_out = n_out;
    args.nslot = SL_STACK + 4 * BFHIP_MAX_TREEDEPTH;
    args.rng = rng;
    args.sc = sc;
    args.vec = vec;
    args.samples = samples;
    args.stats = stats;
    args.n_leapfrog = n_leapfrog;
    args.dbg = g_dbg_buf;
    args.dbg_chain = g_dbg_chain;
    args.dbg_cap = g_dbg_cap;
    const size_t need = (size_t)((n_chain + 15) / 16 * 16) * args.nslot * m.DP * sizeof(double);
    if (ctx->scratch_bytes < need) {  // grow-only workspace; allocation is outside any timed region after the first call
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) BF_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = NULL;
        ctx->scratch_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    args.scratch = (double *)ctx->scratch;
    const bool nuts = cfg->sampler == 0;
    switch (W) {
    case 1: return nuts ? launch_sampler<1, true>(ctx, args) : launch_sampler<1, false>(ctx, args);
    case 2: return nuts ? launch_sampler<2, true>(ctx, args) : launch_sampler<2, false>(ctx, args);
    case 4: return nuts ? launch_sampler<4, true>(ctx, args) : launch_sampler<4, false>(ctx, args);
    case 8: return nuts ? launch_sampler<8, true>(ctx, args) : launch_sampler<8, false>(ctx, args);
    }
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "unsupported padded dimension %d", m.DP);
}
