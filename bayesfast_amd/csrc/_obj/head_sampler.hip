// bfhip_sampler.hip -- fused NUTS / HMC transitions for many chains (gfx950).
//
// Work decomposition ("tile phase / chain phase"):
//   * A workgroup of 16 wavefronts owns a GROUP of 16 chains for the whole launch: 4 waves per SIMD, so the
//     latency-bound per-chain logic of one wave hides behind the other three.
//   * Chain phase (everything O(d)): ONE WAVE PER CHAIN.  Lane l holds dimensions l*E .. l*E+E-1 of every
//     state vector (E = DP/64, 1 at d <= 64).  All tree control flow is wave-uniform: no cross-chain
//     divergence, dot products are wave reductions, per-chain scalars live once per wave.
//   * Tile phase (gradient): the batched matvecs G^T = S X^T and H (X - mu)^T of the 16 chains run on
//     v_mfma_f64_16x16x4_f64 in the first W = DP/16 waves (wave w: output rows 16w..16w+15 for all 16 chains).
//     The coefficient matrices S, H (, H_decay) are staged once per launch in LDS as A-operand fragments.
//   * The two layouts meet in LDS: XB (B operands, written by the chain waves) and GB (matvec results).
//     Two workgroup barriers per trip, none inside the tree logic.
//
// Every chain is an independent state machine (INIT -> LEAF ... -> iteration end -> INIT ...); one loop
// trip evaluates ONE gradient for all 16 chains of the group, whatever each chain needs it for.  The
// compute_state() call that opens every iteration (base_hmc.py:70) is a leapfrog with epsilon = 0, so
// chains never wait for each other inside an iteration or across iterations.
//
// The recursion of Tree._build_subtree (samplers/nuts.py:134-178) is flattened: leaf i of a 2^depth
// subtree is merged upwards while bit `level` of i is set; completed sub-subtrees wait on a per-chain
// stack (vectors in global scratch, scalars in LDS).  The per-chain logic is time-sliced into UNITS (finish
// an evaluation / one merge level / end of a doubling / three pieces of the iteration end), one unit per
// chain per trip, so the workgroup barrier never waits for a long bookkeeping path of one chain.  Random draws are consumed in the recursion's
// post-order, so a chain reproduces the CPU oracle's trajectory for the same xoshiro stream.
#include "bfhip_eval.h"

// The f64 libm expansions (exp, log, sincospi, sqrt) are long inline sequences whose constants get hoisted
// out of the trip loop; inlined at every call site they push the kernel far over its 128-VGPR budget
// (68 spilled VGPRs, 228 B of scratch per lane, reloaded inside the hot loop).  Out-of-line copies keep the
// hot loop's register pressure down (32 spilled); the calls are rare (a few per chain per trip).
__device__ __attribute__((noinline)) static double bf_exp_ni(double x) { return exp(x); }
__device__ __attribute__((noinline)) static double bf_log_ni(double x) { return log(x); }
__device__ __attribute__((noinline)) static double bf_sqrt_ni(double x) { return sqrt(x); }
__device__ __attribute__((noinline)) static void bf_sincospi_ni(double x, double *s, double *c) { sincospi(x, s, c); }
#define exp(x) bf_exp_ni(x)
#define log(x) bf_log_ni(x)
#define sqrt(x) bf_sqrt_ni(x)
#define sincospi(x, s, c) bf_sincospi_ni(x, s, c)

struct SamplerArgs {
    bfhip_sampler_config cfg;
    int n_chain, iter_end, iter_out0, n_out, nslot;
    uint64_t *rng;
    double *sc, *vec, *samples, *stats;
    unsigned long long *n_leapfrog;
    double *scratch;
    unsigned long long *stamps;  // diagnostics only: [groups][16 waves][20]: 10 cycle counters + 10 event counts, or NULL
};

enum { M_INIT = 0, M_LEAF = 1, M_OOB = 2, M_DONE = 3 };
// units of per-chain work; a chain runs one per trip (U_EVAL needs this trip's gradient)
enum { U_EVAL = 0, U_MERGE_RUN, U_DBL_END, U_END1, U_END2, U_END3, U_DONE, U_MERGE, U_ABORT };
enum { SL_LEFT_Q = 0, SL_LEFT_P, SL_LEFT_G, SL_RIGHT_Q, SL_RIGHT_P, SL_RIGHT_G, SL_PROP_Q, SL_PSUM, SL_STACK };
enum { LS_LS = 0, LS_E, LS_LOGP, LS_ACC, LS_N };
// cold per-chain scalars parked in LDS (one writer: lane 0 of the chain's wave; broadcast reads)
enum { CS_LOG_STEP = 0, CS_LOG_BAR, CS_HBAR, CS_SMU, CS_COUNT, CS_PROP_E, CS_PROP_LOGP, CS_MAX_DE, CS_HACC, CS_HDE,
       CS_W_OFF, CS_TREE_W, CS_BETA, CS_T_E, CS_T_LOGP, CS_N };

// one DPP move of a double (both halves)
template <int CTRL>
__device__ inline double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// sum over the 64 lanes, wave-uniform result: DPP butterflies inside each row of 16 lanes (no LDS
// crossbar), then the four row totals are read to scalar registers and added in a fixed order
__device__ inline double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror
    return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}

template <int W>
struct SamplerGeo {
    static constexpr int DP = 16 * W, NS = 4 * W;
    static constexpr int E = DP >= 64 ? DP / 64 : 1;  // state elements per lane
    static constexpr int XS = 65;                     // XB row stride (doubles)
    static constexpr int GS = DP + 1;                 // GB row stride
    static constexpr int MAT = DP * DP;
    static constexpr bool STAGE = DP <= 64;           // coefficient fragments fit in LDS
};

template <int W, bool NUTS, bool STAMPS>
__global__ __launch_bounds__(1024) void bf_sampler_kernel(DevModel m, SamplerArgs a) {
    using G = SamplerGeo<W>;
    constexpr int DP = G::DP, NS = G::NS, E = G::E, XS = G::XS, GS = G::GS, MAT = G::MAT;
    constexpr int MAXL = BFHIP_MAX_TREEDEPTH;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *XB = lds;                         // [3][NS][XS]   B operands
    double *GB = XB + 3 * NS * XS;            // [3][16][GS]   matvec results
    double *LS = GB + 3 * 16 * GS;            // [16][MAXL][LS_N] per-chain stack scalars
    int *alive = (int *)(LS + 16 * MAXL * LS_N);  // [2] (+ pad)
    double *CS = LS + 16 * MAXL * LS_N + 2;   // [16][CS_N]    cold per-chain scalars (kept out of the VGPR budget)
    double *PDL = CS + 16 * CS_N;             // [PD_N][DP]    per-dimension table (rarely used rows are read from here)
    double *FR = PDL + PD_N * DP;             // staged A fragments: S | H | H_decay

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index == chain index in the group
    const int chain = blockIdx.x * 16 + w;
    const bool real = chain < a.n_chain;
    const int d = m.d;

    // ---- stage the coefficient matrices (A-operand fragments) in LDS ----
    const double *Sf = m.Sf, *Hf = m.Hf, *Hdf = m.Hdf;
    if constexpr (G::STAGE) {
        double *pS = FR, *pH = pS + (m.has_quad ? MAT : 0), *pD = pH + (m.use_bound ? MAT : 0);
        if (m.has_quad)
            for (int i = tid; i < MAT / 2; i += 1024) ((d2_t *)pS)[i] = ((const d2_t *)m.Sf)[i];
        if (m.use_bound)
            for (int i = tid; i < MAT / 2; i += 1024) ((d2_t *)pH)[i] = ((const d2_t *)m.Hf)[i];
        if (m.use_decay)
            for (int i = tid; i < MAT / 2; i += 1024) ((d2_t *)pD)[i] = ((const d2_t *)m.Hdf)[i];
        Sf = pS; Hf = pH; Hdf = pD;
    }
    if (tid < 2) alive[tid] = 0;
    for (int i = tid; i < PD_N * DP; i += 1024) PDL[i] = m.pd[i];

    // ---- per-lane constants: the per-dimension table rows of this lane's dimensions ----
    double c_lin[E], c_mu[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int dim = lane * E + e;
        const bool in = dim < DP;
        c_lin[e] = in ? m.pd[PD_LIN * DP + dim] : 0.;
        c_mu[e] = in ? m.pd[PD_MU * DP + dim] : 0.;
    }
    // rows of the per-dimension table that only optional features read (from LDS, in their branches)
    auto pdl = [&](int rowi, int e) -> double {
        const int dim = lane * E + e;
        return dim < DP ? PDL[rowi * DP + dim] : ((rowi == PD_RG || rowi == PD_SU_DIFF) ? 1. : 0.);
    };

    // ---- per-chain state (scalars are wave-uniform) ----
    double q[E], p[E], g[E], var[E], TLp[E], TPs[E], TPq[E];
    double L0p[E], L0q[E];           // stack level 0 (a single waiting leaf): its p and q
    double PF0[E], PF1[E], PF2[E], PF3[E];  // vectors the NEXT unit needs, loaded one trip ahead (latency hides in the barrier)
    uint64_t rs[4] = {0, 0, 0, 0};
    int i_iter = 0, mode = M_DONE, prev_mode = M_INIT, err = 0;
    double eps = 0., eps_t = 0.;
    int dir = 1, depth = 0, i_leaf = 0, n_prop = 0, diverged = 0;
    double start_energy = 0., acc_sum = 0.;
    double T_W = 0., T_acc = 0.;
    int unit = U_DONE, lev = 0, h_accepted = 0;
    double *csw = CS + w * CS_N;
    auto cs_set = [&](int i, double v) { if (lane == 0) csw[i] = v; };
    unsigned long long nlf = 0;
    const bool lane_ok = lane * E < DP;  // lanes beyond the padded dimension idle (DP < 64)
    double *sbase = a.scratch + ((size_t)(real ? chain : 0) * a.nslot) * DP + lane * E;
    double *scp = a.sc + (size_t)(real ? chain : 0) * BFHIP_SC_N;
    double *vecp = a.vec + (size_t)(real ? chain : 0) * BFHIP_VEC_N * d;
    double *lsw = LS + w * (MAXL * LS_N);
    const int nw = a.cfg.n_warmup;

    auto load_vec = [&](int field, double (&v)[E], double pad) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = lane * E + e;
            v[e] = (dim < d) ? vecp[field * d + dim] : pad;
        }
    };
    auto store_vec = [&](int field, const double (&v)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = lane * E + e;
            if (dim < d) vecp[field * d + dim] = v[e];
        }
    };
    auto ldv = [&](int slot, double (&v)[E]) {  // scratch vector slot -> registers (coalesced, 512 B per wave at E = 1)
        if (lane_ok) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = sbase[(size_t)slot * DP + e];
        }
    };
    auto stv = [&](int slot, const double (&v)[E]) {
        if (lane_ok) {
#pragma unroll
            for (int e = 0; e < E; ++e) sbase[(size_t)slot * DP + e] = v[e];
        }
    };
    // metric.random: samplers/hmc_utils/metrics.py:83-86.  One xoshiro draw K keys a SplitMix64 counter
    // stream; pair P of the stream gives dimensions 2P (cos) and 2P+1 (sin) by Box-Muller.
    auto draw_momentum = [&]() {
        const uint64_t K = bf_xoshiro_next(rs);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = lane * E + e;
            const uint64_t P = (uint64_t)(dim >> 1);
            const double u1 = bf_u01_open0(bf_mix64(K + (2 * P + 1) * BF_GOLDEN));
            const double u2 = bf_u01(bf_mix64(K + (2 * P + 2) * BF_GOLDEN));
            const double rad = sqrt(-2. * log(u1));
            double sn, cs;
            sincospi(2. * u2, &sn, &cs);  // angle 2 pi u2 without a large-argument reduction
            const double z = (dim & 1) ? rad * sn : rad * cs;
            p[e] = (dim < d) ? (1. / sqrt(var[e])) * z : 0.;
            g[e] = 0.;
        }
    };
#pragma unroll
    for (int e = 0; e < E; ++e) { q[e] = 0.; p[e] = 0.; g[e] = 0.; var[e] = 1.; TLp[e] = 0.; TPs[e] = 0.; TPq[e] = 0.; PF0[e] = 0.; PF1[e] = 0.; PF2[e] = 0.; PF3[e] = 0.; L0p[e] = 0.; L0q[e] = 0.; }
    if (real) {
        for (int k = 0; k < 4; ++k) rs[k] = a.rng[(size_t)chain * 4 + k];
        cs_set(CS_LOG_STEP, scp[BFHIP_SC_LOG_STEP]);
        cs_set(CS_LOG_BAR, scp[BFHIP_SC_LOG_BAR]);
        cs_set(CS_HBAR, scp[BFHIP_SC_HBAR]);
        cs_set(CS_SMU, scp[BFHIP_SC_MU]);
        cs_set(CS_COUNT, scp[BFHIP_SC_COUNT]);
        i_iter = (int)scp[BFHIP_SC_I_ITER];
        err = (int)scp[BFHIP_SC_ERROR];
        load_vec(BFHIP_VEC_Q, q, 0.);
        load_vec(BFHIP_VEC_VAR, var, 1.);
        if (i_iter < a.iter_end && err == 0) {
            mode = M_INIT;
            unit = U_EVAL;
            draw_momentum();
        }
    }
    __syncthreads();

    // ---- the per-chain state machine: ONE unit of work per call (wave-uniform control flow) ----
    auto run_unit = [&](bool have_ev, double E_new, double logp_new) {
        // ================= per-chain state machine: ONE unit of work per trip =================
        // (wave-uniform control flow; the barrier-to-barrier critical path is the longest single unit)
        if (unit == U_EVAL) {
            if (have_ev && mode == M_INIT) {
                // BaseHMC.astep start: base_hmc.py:70-76, Tree.__init__: nuts.py:24-43
                if (!(fabs(E_new) <= 1.7976931348623157e308)) {
                    err = 1;
                } else {
                    start_energy = E_new;
                    stv(SL_LEFT_Q, q); stv(SL_LEFT_P, p); stv(SL_LEFT_G, g);
                    stv(SL_RIGHT_Q, q); stv(SL_RIGHT_P, p); stv(SL_RIGHT_G, g);
                    stv(SL_PROP_Q, q); stv(SL_PSUM, p);
                    cs_set(CS_PROP_E, E_new);
                    cs_set(CS_PROP_LOGP, logp_new);
                    cs_set(CS_TREE_W, 1.);
                    cs_set(CS_W_OFF, 0.);
                    cs_set(CS_MAX_DE, 0.);
                    depth = 0; acc_sum = 0.; n_prop = 0; diverged = 0; i_leaf = 0;
                    eps = exp(i_iter < nw ? csw[CS_LOG_STEP] : csw[CS_LOG_BAR]);  // step_size.py:25-29
                    dir = 1;
                    if (NUTS) dir = (bf_u01(bf_xoshiro_next(rs)) < 0.5) ? 1 : -1;  // nuts.py:210, log(U) < log(1/2)
                    mode = M_LEAF;
                }
            } else if (have_ev && mode == M_LEAF) {
                nlf += 1;
                if (NUTS) {
                    // ---- Tree._single_step: nuts.py:105-132 ----
                    n_prop += 1;
                    double dE = E_new - start_energy;
                    if (dE != dE) dE = INFINITY;
                    if (fabs(dE) > fabs(csw[CS_MAX_DE])) cs_set(CS_MAX_DE, dE);
                    cs_set(CS_T_E, E_new);
                    cs_set(CS_T_LOGP, logp_new);
                    T_acc = 0.; lev = 0;
                    if (fabs(dE) < a.cfg.max_change) {
                        // multinomial weight exp(log_size) = exp(-dE), kept in the linear domain relative to
                        // a running offset w_off (exact streaming log-sum-exp; rescales are rare)
                        const double w_off = csw[CS_W_OFF];
                        double aw = -dE - w_off;
                        if (aw > 600.) {
                            const double sc_ = exp(-aw);
                            cs_set(CS_TREE_W, csw[CS_TREE_W] * sc_);
                            if (lane == 0)
                                for (int l2 = 0; l2 < depth; ++l2) lsw[l2 * LS_N + LS_LS] *= sc_;
                            cs_set(CS_W_OFF, w_off + aw);
                            aw = 0.;
                        }
                        T_W = exp(aw);
                        const double pacc = (csw[CS_W_OFF] == 0.) ? T_W : exp(-dE);
                        T_acc = pacc > 1. ? 1. : pacc;
#pragma unroll
                        for (int e = 0; e < E; ++e) { TLp[e] = p[e]; TPs[e] = p[e]; TPq[e] = q[e]; }
                        unit = U_MERGE;  // resolved below (push / complete need no further trip)
                        if ((i_leaf & 1) && depth > 0) {
                            // ---- level-0 merge with the previous leaf, whose (p, q) wait in L0p / L0q ----
                            // (single leaves: left.p = right.p = p_sum, and no extra checks at depth 1, nuts.py:154)
                            double d0 = 0., d1 = 0.;
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const double ps0 = L0p[e] + p[e];
                                d0 += ps0 * (var[e] * L0p[e]);  // nuts.py:150-151
                                d1 += ps0 * (var[e] * p[e]);
                            }
                            d0 = wave_sum(d0);
                            d1 = wave_sum(d1);
                            T_acc = lsw[LS_ACC] + T_acc;  // :173
                            const double Wsum = lsw[LS_LS] + T_W;
                            if (Wsum != Wsum) err = 2;
                            const double u = bf_u01(bf_xoshiro_next(rs));  // :163-167, drawn even when turning
                            if ((d0 <= 0.) || (d1 <= 0.)) {
                                unit = U_ABORT;
                                lev = 1;
                            } else {
                                if (!((u * Wsum < T_W) || (u == 0.))) {
#pragma unroll
                                    for (int e = 0; e < E; ++e) TPq[e] = L0q[e];
                                    cs_set(CS_T_E, lsw[LS_E]);
                                    cs_set(CS_T_LOGP, lsw[LS_LOGP]);
                                }
                                T_W = Wsum;
#pragma unroll
                                for (int e = 0; e < E; ++e) { TPs[e] = L0p[e] + p[e]; TLp[e] = L0p[e]; }
                                lev = 1;
                            }
                        }
                    } else {
                        diverged = 1;
                        unit = U_ABORT;
                    }
                } else {
                    // ---- HMC._hamiltonian_step: samplers/hmc.py:16-49 ----
                    i_leaf += 1;
                    if (i_leaf >= a.cfg.n_int_step) {
                        const bool fin = fabs(E_new) <= 1.7976931348623157e308;
                        const double h_dE = fin ? (start_energy - E_new) : -INFINITY;
                        diverged = (!fin || fabs(h_dE) > a.cfg.max_change) ? 1 : 0;
                        double h_accept_stat = exp(h_dE);
                        if (h_accept_stat > 1.) h_accept_stat = 1.;
                        h_accepted = 0;
                        if (!diverged) h_accepted = !(bf_u01(bf_xoshiro_next(rs)) >= h_accept_stat);
                        if (h_accepted) stv(SL_PROP_Q, q);
                        cs_set(CS_HDE, h_dE);
                        cs_set(CS_HACC, h_accept_stat);
                        cs_set(CS_PROP_E, E_new);
                        cs_set(CS_PROP_LOGP, logp_new);
                        unit = U_END1;
                    }
                }
            }
        } else if (unit == U_MERGE_RUN) {
            // ---- one level of Tree._build_subtree's merge (nuts.py:146-178) ----
            double A[E], B[E], S1[E], psum[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { A[e] = PF0[e]; B[e] = PF1[e]; S1[e] = PF2[e]; }  // prefetched when this unit was scheduled
            double d0 = 0., d1 = 0., d2 = 0., d3 = 0., d4 = 0., d5 = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                psum[e] = S1[e] + TPs[e];
                const double vA = var[e] * A[e], vB = var[e] * B[e], vC = var[e] * TLp[e], vD = var[e] * p[e];
                d0 += psum[e] * vA;  // nuts.py:150-151
                d1 += psum[e] * vD;
                const double ps1 = S1[e] + TLp[e];  // :155-157
                d2 += ps1 * vA;
                d3 += ps1 * vC;
                const double ps2 = B[e] + TPs[e];   // :158-160
                d4 += ps2 * vB;
                d5 += ps2 * vD;
            }
            d0 = wave_sum(d0);
            d1 = wave_sum(d1);
            bool turning = (d0 <= 0.) || (d1 <= 0.);
            if (lev >= 1) {
                d2 = wave_sum(d2); d3 = wave_sum(d3); d4 = wave_sum(d4); d5 = wave_sum(d5);
                turning = turning || (d2 <= 0.) || (d3 <= 0.) || (d4 <= 0.) || (d5 <= 0.);
            }
            const double *lsp = lsw + lev * LS_N;
            T_acc = lsp[LS_ACC] + T_acc;  // :173
            // nuts.py:163-167 run even when THIS merge's check says turning: the draw is consumed.
            // logbern(ls2 - logaddexp(ls1, ls2))  <=>  U * (W1 + W2) < W2
            const double Wsum = lsp[LS_LS] + T_W;
            if (Wsum != Wsum) err = 2;
            const double u = bf_u01(bf_xoshiro_next(rs));
            const bool keep_t2 = (u * Wsum < T_W) || (u == 0.);
            if (turning) {
                unit = U_ABORT;
                lev += 1;  // ancestors above this level still add their accept sums
            } else {
                if (!keep_t2) {
#pragma unroll
                    for (int e = 0; e < E; ++e) TPq[e] = PF3[e];  // the sibling's proposal, prefetched
                    cs_set(CS_T_E, lsp[LS_E]);
                    cs_set(CS_T_LOGP, lsp[LS_LOGP]);
                }
                T_W = Wsum;
#pragma unroll
                for (int e = 0; e < E; ++e) { TLp[e] = A[e]; TPs[e] = psum[e]; }
                lev += 1;
                unit = U_MERGE;
            }
        } else if (unit == U_DBL_END) {
            // ---- Tree.extend after a complete subtree: nuts.py:71-103 ----
            double oldL[E], oldR[E], ps[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { ps[e] = PF0[e]; oldL[e] = PF1[e]; oldR[e] = PF2[e]; }  // prefetched
            depth += 1;
            acc_sum += T_acc;
            {   // :81-83  logbern(ls_new - ls_old)  <=>  U * W_old < W_new
                const double tree_W = csw[CS_TREE_W];
                if (T_W != T_W || tree_W != tree_W) err = 2;
                const double u = bf_u01(bf_xoshiro_next(rs));
                if ((u * tree_W < T_W) || (u == 0.)) {
                    stv(SL_PROP_Q, TPq);
                    cs_set(CS_PROP_E, csw[CS_T_E]);
                    cs_set(CS_PROP_LOGP, csw[CS_T_LOGP]);
                }
                cs_set(CS_TREE_W, tree_W + T_W);  // :85
            }
            double d0 = 0., d1 = 0., d2 = 0., d3 = 0., d4 = 0., d5 = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ps[e] += TPs[e];  // :86 (in place)
                const double vN = var[e] * p[e], vT = var[e] * TLp[e], vL = var[e] * oldL[e], vR = var[e] * oldR[e];
                // NOTE (reference behaviour, kept on purpose): leftmost_p_sum (dir > 0) / rightmost_p_sum
                // (dir < 0) alias self.p_sum, which line 86 has just updated in place.
                if (dir > 0) {
                    d0 += ps[e] * vL;                     // left = old left
                    d1 += ps[e] * vN;                     // right = new end
                    const double ps1 = ps[e] + TLp[e];    // (aliased) leftmost_p_sum + rightmost_begin.p
                    d2 += ps1 * vL;                       // leftmost_begin = old left
                    d3 += ps1 * vT;                       // rightmost_begin = tree.left
                    const double ps2 = oldR[e] + TPs[e];  // leftmost_end.p + rightmost_p_sum
                    d4 += ps2 * vR;                       // leftmost_end = old right
                    d5 += ps2 * vN;                       // rightmost_end = tree.right
                } else {
                    d0 += ps[e] * vN;                     // left = new end
                    d1 += ps[e] * vR;                     // right = old right
                    const double ps1 = TPs[e] + oldL[e];  // leftmost_p_sum + rightmost_begin.p
                    d2 += ps1 * vN;                       // leftmost_begin = tree.right
                    d3 += ps1 * vL;                       // rightmost_begin = old left
                    const double ps2 = TLp[e] + ps[e];    // leftmost_end.p + (aliased) rightmost_p_sum
                    d4 += ps2 * vT;                       // leftmost_end = tree.left
                    d5 += ps2 * vR;                       // rightmost_end = old right
                }
            }
            d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
            d3 = wave_sum(d3); d4 = wave_sum(d4); d5 = wave_sum(d5);
            stv(SL_PSUM, ps);
            const int eo = (dir > 0) ? SL_RIGHT_Q : SL_LEFT_Q;
            stv(eo + 0, q); stv(eo + 1, p); stv(eo + 2, g);
            const bool turning = (d0 <= 0.) || (d1 <= 0.) || (d2 <= 0.) || (d3 <= 0.) || (d4 <= 0.) || (d5 <= 0.);
            if (turning || depth >= a.cfg.max_treedepth) {
                unit = U_END1;
            } else {
                const int nd = (bf_u01(bf_xoshiro_next(rs)) < 0.5) ? 1 : -1;  // nuts.py:210
                if (nd != dir) {
                    const int eo2 = (nd > 0) ? SL_RIGHT_Q : SL_LEFT_Q;
                    ldv(eo2 + 0, q); ldv(eo2 + 1, p); ldv(eo2 + 2, g);
                }
                dir = nd;
                i_leaf = 0;
                unit = U_EVAL;
            }
        } else if (unit == U_END1) {
            // ================= iteration end, part 1: step size + stats (base_hmc.py:80-85) =================
            const bool warm = i_iter < nw;
            const double accept_stat = NUTS ? acc_sum / (double)n_prop : csw[CS_HACC];  // nuts.py:186
            double log_step = csw[CS_LOG_STEP], log_bar = csw[CS_LOG_BAR];
            if (warm && a.cfg.adapt_step_size) {  // step_size.py:31-45
                const double count = csw[CS_COUNT];
                const double wgt = 1. / (count + a.cfg.t_0);
                const double hbar = ((1. - wgt) * csw[CS_HBAR] + wgt * (a.cfg.target_accept - accept_stat));
                log_step = csw[CS_SMU] - hbar * sqrt(count) / a.cfg.gamma;
                const double mk = exp(-a.cfg.k * log(count));  // count ** -k
                log_bar = mk * log_step + (1. - mk) * log_bar;
                cs_set(CS_HBAR, hbar);
                cs_set(CS_LOG_STEP, log_step);
                cs_set(CS_LOG_BAR, log_bar);
                cs_set(CS_COUNT, count + 1.);
            }
            const double prop_E = csw[CS_PROP_E], prop_logp = csw[CS_PROP_LOGP];
            const int orow = i_iter - a.iter_out0;
            if (orow >= 0 && orow < a.n_out && lane == 0) {
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
                    st[BFHIP_NS_MAX_ENERGY_CHANGE] = csw[CS_MAX_DE];
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
                    st[BFHIP_HS_ENERGY_CHANGE] = csw[CS_HDE];
                    st[BFHIP_HS_DIVERGING] = (double)diverged;
                    st[10] = 0.;
                }
            }
            ldv(SL_PROP_Q, PF0);
            unit = U_END2;
        } else if (unit == U_END2) {
            // ================= iteration end, part 2: the new sample + metric adaptation =================
            const bool warm = i_iter < nw;
            const int orow = i_iter - a.iter_out0;
#pragma unroll
            for (int e = 0; e < E; ++e) q[e] = PF0[e];  // the proposal, prefetched by part 1
            if (orow >= 0 && orow < a.n_out) {
                double *sp = a.samples + ((size_t)chain * a.n_out + orow) * d;
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (lane * E + e < d) sp[lane * E + e] = q[e];
            }
            // QuadMetricDiagAdapt.update: metrics.py:186-211, _WeightedVariance.add_sample :354-360
            if (warm && a.cfg.adapt_metric) {
                double fg_n = scp[BFHIP_SC_FG_N], bg_n = scp[BFHIP_SC_BG_N];
                double n_samples = scp[BFHIP_SC_N_SAMPLES], prev_upd = scp[BFHIP_SC_PREV_UPDATE];
                double adapt_window = scp[BFHIP_SC_ADAPT_WINDOW];
                const long delta = (long)(n_samples - prev_upd);
                double fm[E], fr[E], bm[E], br[E];
                load_vec(BFHIP_VEC_FG_MEAN, fm, 0.);
                load_vec(BFHIP_VEC_FG_RAW, fr, 0.);
                load_vec(BFHIP_VEC_BG_MEAN, bm, 0.);
                load_vec(BFHIP_VEC_BG_RAW, br, 0.);
                fg_n += 1.;
                bg_n += 1.;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double od = q[e] - fm[e];
                    fm[e] += od / fg_n;
                    fr[e] += 1. * od * (q[e] - fm[e]);
                    od = q[e] - bm[e];
                    bm[e] += od / bg_n;
                    br[e] += 1. * od * (q[e] - bm[e]);
                }
                if ((delta + 1) % (long)a.cfg.update_window == 0) {  // metrics.py:181-184
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (lane * E + e < d) var[e] = fr[e] / fg_n;
                    store_vec(BFHIP_VEC_VAR, var);
                }
                if ((double)delta >= adapt_window) {
#pragma unroll
                    for (int e = 0; e < E; ++e) { fm[e] = bm[e]; fr[e] = br[e]; bm[e] = 0.; br[e] = 0.; }
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
                if (lane == 0) {
                    scp[BFHIP_SC_FG_N] = fg_n;
                    scp[BFHIP_SC_BG_N] = bg_n;
                    scp[BFHIP_SC_N_SAMPLES] = n_samples;
                    scp[BFHIP_SC_PREV_UPDATE] = prev_upd;
                    scp[BFHIP_SC_ADAPT_WINDOW] = adapt_window;
                }
            }
            i_iter += 1;
            unit = U_END3;
        } else if (unit == U_END3) {
            // ================= iteration end, part 3: next momentum =================
            if (i_iter < a.iter_end && err == 0) {
                mode = M_INIT;
                draw_momentum();
                unit = U_EVAL;
            } else {
                mode = M_DONE;
                unit = U_DONE;
            }
        }
        // ---- cheap follow-ups that need no trip of their own ----
        if (unit == U_ABORT) {
            // unwind: every pending ancestor adds its left half's accept_sum (nuts.py:173)
            for (int al = (diverged ? 0 : lev); al < depth; ++al)
                if ((i_leaf >> al) & 1) T_acc = lsw[al * LS_N + LS_ACC] + T_acc;
            depth += 1;  // nuts.py:71-73
            acc_sum += T_acc;
            unit = U_END1;
        } else if (unit == U_MERGE) {
            if (lev < depth && ((i_leaf >> lev) & 1)) {
                unit = U_MERGE_RUN;  // next trip: merge with the waiting left sibling at this level
                ldv(SL_STACK + 4 * lev + 0, PF0);
                ldv(SL_STACK + 4 * lev + 1, PF1);
                ldv(SL_STACK + 4 * lev + 2, PF2);
                ldv(SL_STACK + 4 * lev + 3, PF3);
            } else if (lev < depth) {
                // the subtree waits for its right sibling
                if (lev == 0) {
#pragma unroll
                    for (int e = 0; e < E; ++e) { L0p[e] = p[e]; L0q[e] = q[e]; }  // stack level 0 lives in registers
                } else {
                    const int slot = SL_STACK + 4 * lev;
                    stv(slot + 0, TLp); stv(slot + 1, p); stv(slot + 2, TPs); stv(slot + 3, TPq);
                }
                if (lane == 0) {
                    double *lsp = lsw + lev * LS_N;
                    lsp[LS_LS] = T_W; lsp[LS_E] = csw[CS_T_E]; lsp[LS_LOGP] = csw[CS_T_LOGP]; lsp[LS_ACC] = T_acc;
                }
                i_leaf += 1;
                unit = U_EVAL;
            } else {
                unit = U_DBL_END;
                ldv(SL_PSUM, PF0);
                ldv(SL_LEFT_P, PF1);
                ldv(SL_RIGHT_P, PF2);
            }
        }
        if (err != 0) { mode = M_DONE; unit = U_DONE; }
    };

    int mat_id[3] = {0, 0, 0};
    int n_mat = 0;
    if (m.has_quad) mat_id[n_mat++] = 0;
    if (m.use_bound) mat_id[n_mat++] = 1;
    if (m.use_decay) mat_id[n_mat++] = 2;

    // phase stamps exist only in the diagnostic instantiation (they cost 18 always-live VGPRs)
    unsigned long long st_acc[STAMPS ? 10 : 1] = {0}, st_cnt[STAMPS ? 10 : 1] = {0}, st_prev = STAMPS ? clock64() : 0;
    auto stamp = [&](int k) {
        if constexpr (STAMPS) {
            const unsigned long long t = clock64();
            st_acc[k] += t - st_prev;
            st_cnt[k] += 1;
            st_prev = t;
        }
    };

    for (int trip = 0;; ++trip) {
        // ================= phase A: first half of the leapfrog, B operands =================
        double xs[E], jac[E], gj[E], xo[E];
        double logdet = 0.;
        const bool evaluating = unit == U_EVAL;
        if (evaluating) {
            if (mode != M_OOB) {
                eps_t = (mode == M_LEAF) ? eps * (double)dir : 0.;
                const double dt = 0.5 * eps_t;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    p[e] = p[e] + dt * g[e];               // integration.py:80
                    q[e] = q[e] + eps_t * (var[e] * p[e]); // :82-85
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dim = lane * E + e;
                xo[e] = q[e];
                jac[e] = 1.;
                gj[e] = 0.;
                if (m.has_transform) {
                    double J, J2;
                    bf_to_original(q[e], (int)pdl(PD_KIND, e), pdl(PD_LO, e), pdl(PD_RG, e), xo[e], J, J2);
                    logdet += log(fabs(J));
                    jac[e] = J;
                    gj[e] = J2 / J;
                }
                xs[e] = m.has_su ? (xo[e] - pdl(PD_SU_LO, e)) / pdl(PD_SU_DIFF, e) : xo[e];
                double x_eval = xs[e];
                if (mode == M_OOB)  // modules/poly.py:482
                    x_eval = (m.alpha * xs[e] + (csw[CS_BETA] - m.alpha) * c_mu[e]) / csw[CS_BETA];
                if (dim < DP) {
                    const int xi = (dim >> 2) * XS + w + 16 * (dim & 3);  // B[k = dim&3][n = chain] of k-step dim>>2
                    XB[0 * NS * XS + xi] = x_eval;
                    if (m.use_bound) XB[1 * NS * XS + xi] = xs[e] - c_mu[e];
                    if (m.use_decay) XB[2 * NS * XS + xi] = xo[e] - pdl(PD_DMU, e);
                }
            }
        }
        if (unit != U_DONE && lane == 0) alive[trip & 1] = 1;
        stamp(0);
        __syncthreads();  // B1
        stamp(1);
        if (alive[trip & 1] == 0) break;  // every chain of the group is done (uniform)
        if (tid == 0) alive[(trip + 1) & 1] = 0;

        // ================= phase B: gradient tiles on MFMA =================
        // job j = (matrix j / W, row tile j % W); jobs are dealt over the 16 waves so that the S and H tiles of
        // one row range run on different waves (the MFMA pipe of each SIMD sees the same total work)
        {
            const int mc = lane & 15, mg = lane >> 4;
            const int n_job = n_mat * W;
            for (int job = w; job < n_job; job += 16) {
                const int slot_m = job / W, t = job % W;       // slot_m-th enabled matrix
                const int b = mat_id[slot_m];                   // 0 S, 1 H, 2 H_decay
                const double *Af = b == 0 ? Sf : (b == 1 ? Hf : Hdf);
                d4_t acc = {0., 0., 0., 0.};
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[(t * NS + s) * 64 + lane], XB[(b * NS + s) * XS + lane], acc, 0, 0, 0);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) GB[(b * 16 + mc) * GS + 16 * t + 4 * r4 + mg] = acc[r4];
            }
        }
        const int unit_in = unit;
        stamp(2);
        __syncthreads();  // B2
        stamp(1);

        // ================= phase C: finish the evaluation =================
        double gn[E], hv[E], dgr[E];
        double logp_new = 0., E_new = 0.;
        bool have_eval = false, kin_ready = false;
        double kin_fast = 0.;
        if (evaluating) {
            double r_quad = 0., r_lin = 0., r_b2 = 0., r_dotj = 0., r_bd2 = 0., r_kin = 0.;
            const bool fast_kin = !m.use_decay && mode != M_OOB;
            double xev[E], r_cub = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dim = lane * E + e;
                const double sx = (m.has_quad && lane_ok) ? GB[(0 * 16 + w) * GS + dim] : 0.;
                hv[e] = (m.use_bound && lane_ok) ? GB[(1 * 16 + w) * GS + dim] : 0.;
                dgr[e] = (m.use_decay && lane_ok) ? GB[(2 * 16 + w) * GS + dim] : 0.;
                xev[e] = xs[e];
                if (mode == M_OOB) xev[e] = (m.alpha * xs[e] + (csw[CS_BETA] - m.alpha) * c_mu[e]) / csw[CS_BETA];
                r_quad += xev[e] * sx;
                r_lin += c_lin[e] * xev[e];
                gn[e] = sx + c_lin[e];
            }
            if (m.has_cubic) {  // cubic configs: x_k of this chain is lane k / E, element k % E
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double gc, fc;
                    bf_cubic_grad(m, lane * E + e, xev[e],
                                  [&](int k) { return readlane_f64((E > 1 && (k % E)) ? xev[E - 1] : xev[0], k / E); }, gc, fc);
                    gn[e] += gc;
                    r_cub += fc;
                }
                r_cub = wave_sum(r_cub);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double xm = xs[e] - c_mu[e];
                r_b2 += xm * hv[e];
                r_dotj += gn[e] * xm;  // dot(jj_0, x - mu), poly.py:496 (used in the OOB pass only)
                if (m.use_decay) r_bd2 += (xo[e] - pdl(PD_DMU, e)) * dgr[e];
                if (fast_kin) {  // in-bound gradient is already final: the kinetic energy rides along
                    double ge = gn[e];
                    if (m.has_su) ge = ge / pdl(PD_SU_DIFF, e);
                    ge = ge * jac[e];
                    if (m.has_transform) ge += gj[e];
                    const double pe = p[e] + (0.5 * eps_t) * ge;
                    r_kin += pe * (var[e] * pe);
                }
            }
            if (fast_kin) r_kin = wave_sum(r_kin);
            r_quad = wave_sum(r_quad);
            r_lin = wave_sum(r_lin);
            if (m.use_bound) r_b2 = wave_sum(r_b2);
            if (m.use_bound && mode == M_OOB) r_dotj = wave_sum(r_dotj);
            if (m.use_decay) r_bd2 = wave_sum(r_bd2);
            if (m.has_transform) logdet = wave_sum(logdet);

            double f = ((m.c0 + r_lin) + 0.5 * r_quad) + r_cub;
            const double beta = sqrt(r_b2);
            bool oob_now = false;
            if (m.use_bound) {
                if (mode == M_OOB) {  // second pass: f, gn currently hold f_0 and jj_0 (poly.py:484-496)
                    const double f0 = f, beta_saved = csw[CS_BETA];
                    f = (beta_saved * f0 - (beta_saved - m.alpha) * m.f_mu) / m.alpha;
                    const double coef = (f0 - m.f_mu) / m.alpha - r_dotj / beta_saved;
#pragma unroll
                    for (int e = 0; e < E; ++e) gn[e] = gn[e] + coef * (hv[e] / beta_saved);
                } else if (beta > m.alpha) {
                    oob_now = true;
                }
            }
            kin_ready = fast_kin && !oob_now;
            kin_fast = r_kin;
            if (oob_now) {
                // outside the alpha-ellipsoid: spend one more trip on the projected point x_0
                cs_set(CS_BETA, beta);
                prev_mode = mode;
                mode = M_OOB;
            } else {
                if (mode == M_OOB) mode = prev_mode;
                // chain rule (module.py:226, density.py:558), decay (:740-746), transform (:747-750)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (m.has_su) gn[e] = gn[e] / pdl(PD_SU_DIFF, e);
                    gn[e] = gn[e] * jac[e];
                }
                if (m.use_decay) {
                    f -= m.decay_gamma * bf_clip0(r_bd2 - m.decay_alpha2);
                    if (r_bd2 > m.decay_alpha2) {
#pragma unroll
                        for (int e = 0; e < E; ++e) gn[e] -= 2. * m.decay_gamma * dgr[e];
                    }
                }
                if (m.has_transform) {
                    f += logdet;
#pragma unroll
                    for (int e = 0; e < E; ++e) gn[e] += gj[e];
                }
                logp_new = f;
                have_eval = true;
            }
        }
        if (have_eval) {
            // second half of the leapfrog and the kinetic energy
            double kin = 0.;
            const double dt = 0.5 * eps_t;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                p[e] = p[e] + dt * gn[e];        // integration.py:90
                kin += p[e] * (var[e] * p[e]);   // metrics.py:88-91
                g[e] = gn[e];
            }
            kin = kin_ready ? kin_fast : wave_sum(kin);
            E_new = 0.5 * kin - logp_new;        // integration.py:92-93
        }

        stamp(3);
        const int mode_in = mode;
        // every chain runs ONE unit here, in parallel: chains that evaluated finish their leaf / init, the
        // others do their pending merge level / doubling end / iteration-end piece
        if (unit_in == U_EVAL) run_unit(have_eval, E_new, logp_new);
        else run_unit(false, 0., 0.);
        stamp(unit_in == U_EVAL ? (mode_in == M_INIT ? 4 : 5) : (unit_in == U_MERGE_RUN ? 6 : (unit_in == U_DBL_END ? 7 : (unit_in == U_DONE ? 9 : 8))));
    }

    if constexpr (STAMPS) {
        if (a.stamps && lane == 0)
            for (int k = 0; k < 10; ++k) {
                a.stamps[((size_t)blockIdx.x * 16 + w) * 20 + k] = st_acc[k];
                a.stamps[((size_t)blockIdx.x * 16 + w) * 20 + 10 + k] = st_cnt[k];
            }
    }
    // ---- write the chain state back ----
    if (real) {
        store_vec(BFHIP_VEC_Q, q);
        if (lane == 0) {
            for (int k = 0; k < 4; ++k) a.rng[(size_t)chain * 4 + k] = rs[k];
            scp[BFHIP_SC_LOG_STEP] = csw[CS_LOG_STEP];
            scp[BFHIP_SC_LOG_BAR] = csw[CS_LOG_BAR];
            scp[BFHIP_SC_HBAR] = csw[CS_HBAR];
            scp[BFHIP_SC_COUNT] = csw[CS_COUNT];
            scp[BFHIP_SC_I_ITER] = (double)i_iter;
            scp[BFHIP_SC_ERROR] = (double)err;
            if (a.n_leapfrog && nlf) atomicAdd(a.n_leapfrog, nlf);
        }
    }
}

static size_t sampler_lds_bytes(const DevModel &m) {
    const int W = m.DP / 16, DP = m.DP, NS = 4 * W;
    size_t dbl = (size_t)3 * NS * 65 + (size_t)3 * 16 * (DP + 1) + (size_t)16 * BFHIP_MAX_TREEDEPTH * LS_N + 2 + (size_t)16 * CS_N + (size_t)PD_N * DP;
    if (DP <= 64) dbl += (size_t)DP * DP * ((m.has_quad ? 1 : 0) + (m.use_bound ? 1 : 0) + (m.use_decay ? 1 : 0));
    return dbl * sizeof(double);
}

template <int W, bool NUTS, bool STAMPS>
static int launch_sampler_t(bfhip_ctx *ctx, const SamplerArgs &args) {
    auto k = bf_sampler_kernel<W, NUTS, STAMPS>;
    const size_t lds = sampler_lds_bytes(ctx->model);
    if (lds > 64 * 1024)
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int groups = (args.n_chain + 15) / 16;
    hipLaunchKernelGGL(k, dim3(groups), dim3(1024), lds, ctx->stream, ctx->model, args);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

static unsigned long long *g_stamps = NULL;
// diagnostics hook (not part of include/bfhip.h): per-wave cycle counters of the sampler kernel's phases
extern "C" void bfhip_debug_stamps(unsigned long long *buf) { g_stamps = buf; }

template <int W, bool NUTS>
static int launch_sampler(bfhip_ctx *ctx, const SamplerArgs &args) {
    if (W == 4 && NUTS && args.stamps) return launch_sampler_t<W, NUTS, (W == 4 && NUTS)>(ctx, args);  // diagnostic build, d <= 64 NUTS only
    return launch_sampler_t<W, NUTS, false>(ctx, args);
}

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
    args.stamps = g_stamps;
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
