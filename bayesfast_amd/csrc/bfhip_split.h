// bfhip_split.h -- NUTS transitions for the plain common surrogate at d <= 64, "split" layout (gfx950).
//
// (The text below describes the W = 4 instantiation, 33 <= d <= 64.  bf_split_body<NUTS_ONLY, W> is the same kernel with W
// integrator and W bookkeeper waves for d <= 16 W: W = 2 at 17 <= d <= 32, W = 1 below, where every wave has a SIMD of its
// own -- the group kernel's d / 16 waves left two or three SIMDs of the CU idle there, profiles/r03s_layout_ab.log.)
//
// The group kernel (bfhip_group.h) runs one wave per SIMD, and a lone wave issues an instruction every 7.7 cycles on
// average whatever it is (profiles/r03_knockouts.log, r03b_debranch_*.log: a trip's time is its instruction count; the
// pipe could take one every 4).  Here a workgroup is EIGHT waves for the same 16 chains, two per SIMD, with DISJOINT work:
//
//   integrator waves 0..3   everything that does not depend on a random draw: the leapfrog step, row tile j of S x (and
//                           H (x - mu) unless the bound proof holds) on MFMA, the evaluation's sums, the U-turn sums of the
//                           subtrees the leaf completes and the momentum part of the subtree stack (left p, right p, p_sum per
//                           level, the tree's ends and p_sum): which subtrees a leaf completes follows from its index alone.
//   bookkeeper waves 4..7   everything that does: the tree's scalars (lane per chain, as in the group kernel), the
//                           multinomial merges and the proposal part of the stack, doublings, iteration end, adaptation,
//                           the momentum draw, output rows.  They run ONE LEAF BEHIND the integrators.
//
// The integrator does not wait for the verdict on leaf n: it goes on with leaf n + 1 of the same subtree, which is what
// happens unless the tree ends.  At the end of a doubling (which it recognises by itself) it needs the next direction: the
// bookkeeper ANNOUNCES it one leaf early, read ahead from the chain's random stream -- the number of draws the pending
// bookkeeping will consume is fixed by the tree's structure (one per merge level, one for the proposal swap, then the
// direction: nuts.py:163-167,81-83,210) as long as the tree goes on, and if it does not the announced doubling is never used.
// Every real command (new iteration, stop) bumps the chain's EPOCH past the announced one, and a leaf computed under an
// epoch the bookkeeper did not reach is dropped: one wasted evaluation per iteration, in a trip the bookkeepers need anyway.
// Every leaf that is used is computed from exactly the state the group kernel computes it from, with the same arithmetic
// in the same order, and the bookkeeper's logic is the group kernel's: samples, statistics, adapted state and random
// streams are BIT-IDENTICAL to bf_group_kernel's (tests/test_gpu_sampler.py, tests/test_group_emu.py).
//
// One trip, two workgroup barriers (three when some chain is outside the bound's proof):
//   integrator: commands | half step, operands | B1 | tiles | [B2a: evaluation scalars] | all sums, momentum stack, leaf -> LDS | B2
//   bookkeeper: leaf n-1 and its sums | B1 | state machine, proposals, commands | [B2a] | B2
// Replaces the same reference code as bfhip_group.h (samplers/nuts.py:21-217, base_hmc.py:62-85, integration.py:68-95,
// modules/poly.py:466-503, step_size.py:10-51, metrics.py:135-237,333-371).
#pragma once
#include "bfhip_group.h"

// W_ = 4: 33 <= d <= 64, eight waves, two per SIMD.  W_ = 2: 17 <= d <= 32, two integrator and two bookkeeper waves -- one per
// SIMD, where the group kernel has two waves and two idle SIMDs.  W_ = 1: d <= 16, one of each.
template <int W_>
struct SplitGeoT {
    static constexpr int W = W_, DP = 16 * W_, NS = 4 * W_, MAXL = BFHIP_MAX_TREEDEPTH, LSS = 5 * MAXL + 1, LSH = 2;
    // vectors in LDS, [slot][dimension][chain]
    //   integrators: stack level 1's momenta (left p, right p, p_sum), the tree's OTHER end (q, p, grad) -- the end being
    //                extended is the integrators' own state
    //   bookkeepers: stack level 1's proposal (q, grad); the start of a new iteration (q, p, grad) for the integrators
    //   integrators -> bookkeepers: the finished leaf (q, grad)
    static constexpr int T_STKM = 0, T_OEND = 3, T_STKP = 6, T_START = 8, T_LEAF = 11, NTV = 13;
    // sums posted by the integrators: the evaluation's and the U-turn checks of the subtrees the leaf completes
    static constexpr int E_KIN = 0, E_VAL = 1, E_B2 = 2, E_A1 = 3, E_A2 = 4, U_M0 = 5, U_LV = 7, U_EXT = U_LV + 6 * LSH, NE = U_EXT + 6;
    static constexpr int NDEEP = 6 * (MAXL - 1 - LSH);
    // per-chain words exchanged between the roles (doubles): command, epoch, signed step, depth | leaf tag, provided logp
    static constexpr int X_CMD = 0, X_EPOCH = 1, X_EPS = 2, X_DEPTH = 3, X_ANN = 4, X_TAG = 5, X_LOGP = 6, X_HASLP = 7, NX = 8;
    // per-chain scalars that only the end of an iteration touches (step-size and metric adaptation), parked in LDS
    static constexpr int NCOLD = 12;   // two copies, read / written alternately by iteration parity
    static constexpr size_t lds_doubles() {
        return (size_t)2 * NS * 64 + 4 * 16 + (size_t)NE * W * 16 + (size_t)W * 16 + (size_t)NTV * DP * 16 + (size_t)16 * LSS +
               (size_t)DP * 16 + (size_t)NX * 16 + 16 + (size_t)2 * NCOLD * 16;
    }
    // per-chain global scratch, in vectors of DP doubles: 5 per stack level 2 .. MAXL-1 (momenta 0-2: integrators, proposal
    // 3-4: bookkeepers), then the sums of the merge levels above LSH
    static constexpr int S_DEEP = 5 * (MAXL - 2);
    static constexpr int scratch_slots() { return S_DEEP + (NDEEP * W + DP - 1) / DP; }
};
using SplitGeo = SplitGeoT<4>;

// commands of the bookkeepers: go on | start an iteration at T_START (signed step, depth 0) | evaluate T_START's point with a
// step of length 0 (the launch's first iteration) | next doubling (signed step, depth) | stop; + 8: reload the metric's variances
// SC_ANN / X_ANN: the signed step of the doubling AFTER the one in flight, read ahead from the random stream (below)
enum { SC_CONT = 0, SC_NEW = 1, SC_INIT = 2, SC_DBL = 3, SC_STOP = 4, SC_ANN = 5 };
// stacked subtree scalars per level: weight (relative to the offset SS_OFF it was stored under), energy and logp of its
// proposal, accept sum
enum { SS_W = 0, SS_E, SS_LOGP, SS_ACC, SS_OFF, SS_N };

// tuning builds (-DBF_STRACE=<n>): lane 0 of the first integrator and of the first bookkeeper wave of workgroup 0 stamp the cycle
// counter at up to 8 points of their first n trips into a.stamps [trip][role][8]
#if defined(BF_STRACE) && !defined(BF_HOST_EMU)
#define STRACE(role, k) do { if (lane == 0 && j == 0 && bf_group() == 0 && a.stamps && trip_no < BF_STRACE) a.stamps[(trip_no * 2 + (role)) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define STRACE(role, k) do { } while (0)
#endif

template <bool NUTS_ONLY = true, int WW = 4>
BF_DEV void bf_split_body(const DevModel &m, const SamplerArgs &a, double *lds) {
    using G = SplitGeoT<WW>;
    constexpr int W = G::W, DP = G::DP, NS = G::NS, LSS = G::LSS;
    double *XB = lds;                          // [2][NS][64]   B operands: x | x - mu
    double *PB = XB + 2 * NS * 64;             // [W][16]       per-wave |x - mu|^2 (bound proof)
    double *RBE = PB + W * 16;                 // [NE][W][16]   sums posted by the integrators
    double *RBK = RBE + G::NE * W * 16;        // [W][16]       the start energy's kinetic part (bookkeepers)
    double *TV = RBK + W * 16;                 // [NTV][DP][16] vectors (SplitGeo)
    double *LS = TV + G::NTV * DP * 16;        // [16][LSS]     subtree stack scalars (bookkeepers)
    double *VARX = LS + 16 * LSS;              // [DP][16]      the metric's variances, bookkeepers -> integrators
    double *XC = VARX + DP * 16;               // [NX][16]      commands / tags
    double *FLG = XC + G::NX * 16;             // [0] some chain is alive
    double *COLD = FLG + 16;                   // [2][NCOLD][16] adaptation scalars (bookkeepers)

    const int tid = bf_tid(), lane = tid & 63, wv = tid >> 6, c = lane & 15, gq = lane >> 4;
    const bool integ = wv < W;
    const int j = integ ? wv : wv - W;
    const int chain = bf_group() * 16 + c;
    const bool real = chain < a.n_chain;
    const int d = m.d, dbase = 16 * j + gq;
    const double bound_thr = m.alpha * m.alpha * (1. - 1e-9);
    double *tvb = TV + dbase * 16 + c;
    double *vxb = VARX + dbase * 16 + c;
    double *sbase = a.scratch + ((size_t)(real ? chain : 0) * a.nslot) * DP + dbase;
    double *rbg = a.scratch + ((size_t)(real ? chain : 0) * a.nslot + G::S_DEEP) * DP;  // this chain's deep sums [v][wave]
    auto tv_ld = [&](int slot, double (&v)[4]) {
        const double *sp = tvb + slot * (DP * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = sp[64 * r];
    };
    auto tv_st = [&](int slot, const double (&v)[4]) {
        double *sp = tvb + slot * (DP * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) sp[64 * r] = v[r];
    };
    // subtree stack vector k of level lev >= 1: 0 left p, 1 right p, 2 p_sum (integrators) | 3 proposal q, 4 proposal gradient
    // (bookkeepers); level 1 in LDS, deeper levels in the context's global scratch
    auto stk_ld = [&](int lev, int k, double (&v)[4]) {
        if (lev == 1) {
            tv_ld(k < 3 ? G::T_STKM + k : G::T_STKP + k - 3, v);
        } else {
            const double *sp = sbase + (size_t)((lev - 2) * 5 + k) * DP;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = sp[4 * r];
        }
    };
    auto stk_st = [&](int lev, int k, const double (&v)[4]) {
        if (lev == 1) {
            tv_st(k < 3 ? G::T_STKM + k : G::T_STKP + k - 3, v);
        } else {
            double *sp = sbase + (size_t)((lev - 2) * 5 + k) * DP;
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[4 * r] = v[r];
        }
    };
    auto sum4 = [](const double (&v)[4]) -> double { return (v[0] + v[1]) + (v[2] + v[3]); };
    auto sumw = [](const double *rp, int st) -> double {   // (the group kernel's association)
        if constexpr (W == 4) return (rp[0] + rp[st]) + (rp[2 * st] + rp[3 * st]);
        else if constexpr (W == 2) return rp[0] + rp[st];
        else return rp[0];
    };
    // the proof's verdict for the point in flight, the same arithmetic in both roles (read after B1)
    auto proof_holds = [&]() -> bool {
        double r2 = PB[c];
#pragma unroll
        for (int w2 = 1; w2 < W; ++w2) r2 += PB[w2 * 16 + c];
        const bool inside = !a.no_bound_proof && m.lam_max * r2 < bound_thr;  // (NaN: not proven)
        return !bf_any(!inside);
    };
    // merge levels of leaf number i_leaf of a doubling of 2^depth leaves: its trailing one bits
    auto merge_levels = [](int i_leaf, int depth) -> int {
        const int t1 = __builtin_ctz(~(unsigned)i_leaf);
        return t1 < depth ? t1 : depth;
    };

#if defined(BF_SPLIT_ONLY) && !defined(BF_HOST_EMU)  // tuning builds: one role's register needs on its own (the kernel hangs)
    if ((BF_SPLIT_ONLY == 1) != integ) return;
#endif
#if defined(BF_K_PRIO) && !defined(BF_HOST_EMU)  // tuning: issue priority of the bookkeepers over the integrators of their SIMD
    if (!integ) __builtin_amdgcn_s_setprio(BF_K_PRIO);
#endif
#if defined(BF_I_PRIO) && !defined(BF_HOST_EMU)
    if (integ) __builtin_amdgcn_s_setprio(BF_I_PRIO);
#endif
    if (integ) {
        // =====================================================================================================
        // INTEGRATOR
        // =====================================================================================================
        double afS[NS], afH[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            afS[s] = m.Sf[(j * NS + s) * 64 + lane];
            afH[s] = m.Hf[(j * NS + s) * 64 + lane];
        }
        double c_lin[4], c_mu[4], c_smu[4], c_hd[4], q[4], p[4], g[4], var[4];
        double L0p[4], PSUM[4], pdbl[4];   // the waiting leaf's momentum, the tree's p_sum, the momentum this doubling started from
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c_lin[r] = m.pd[PD_LIN * DP + dbase + 4 * r];
            c_mu[r] = m.pd[PD_MU * DP + dbase + 4 * r];
            c_smu[r] = m.pd[PD_SMU * DP + dbase + 4 * r];
            c_hd[r] = m.pd[PD_HD * DP + dbase + 4 * r];
            q[r] = p[r] = g[r] = L0p[r] = PSUM[r] = pdbl[r] = 0.;
            var[r] = 1.;
        }
        enum { I_IDLE = 0, I_EVAL = 1 };
        int imode = I_IDLE, epoch_i = 0, i_leaf = 0, depth = 0, dir = 1;
        bool init_eval = false, paused_end = false;   // paused_end: this state is the end of a finished doubling, direction unknown
        double eps_t = 0., ann_eps = 0.;   // ann_eps: the announced next doubling's step (0: none)
        unsigned int n_trip = 0, n_trip_h = 0;
        auto post = [&](int vi, double part) {
            const double t = bf_xor32_add(bf_xor16_add(part));
            if (gq == 0) RBE[(vi * W + j) * 16 + c] = t;
        };
        auto post_n = [&](int vi0, auto &part) {   // as in the group kernel: N sums reduced over the rows side by side
            constexpr int N = sizeof(part) / sizeof(double);
            constexpr int NQ = N / 4, NP = (N % 4) / 2, N1 = N % 2;
            double sq[NQ > 0 ? 2 * NQ : 1], sp[NP > 0 ? NP : 1], s1[N1 > 0 ? N1 : 1];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                sq[2 * k] = bf_pair16_add(part[4 * k], part[4 * k + 1]);
                sq[2 * k + 1] = bf_pair16_add(part[4 * k + 2], part[4 * k + 3]);
            }
            if constexpr (NP > 0) sp[0] = bf_pair16_add(part[4 * NQ], part[4 * NQ + 1]);
            if constexpr (N1 > 0) s1[0] = bf_xor16_add(part[N - 1]);
#pragma unroll
            for (int k = 0; k < NQ; ++k) sq[k] = bf_pair32_add(sq[2 * k], sq[2 * k + 1]);
            if constexpr (NP > 0) sp[0] = bf_xor32_add(sp[0]);
            if constexpr (N1 > 0) s1[0] = bf_xor32_add(s1[0]);
#pragma unroll
            for (int k = 0; k < NQ; ++k) RBE[((vi0 + 4 * k + gq) * W + j) * 16 + c] = sq[k];
            if constexpr (NP > 0) { if (gq < 2) RBE[((vi0 + 4 * NQ + gq) * W + j) * 16 + c] = sp[0]; }
            if constexpr (N1 > 0) { if (gq == 0) RBE[((vi0 + N - 1) * W + j) * 16 + c] = s1[0]; }
        };
        auto post_lv = [&](int lev, int k, double part) {
            if (lev <= G::LSH) {
                post(G::U_LV + 6 * (lev - 1) + k, part);
            } else {
                const double t = bf_xor32_add(bf_xor16_add(part));
                if (gq == 0 && real) rbg[(6 * (lev - 1 - G::LSH) + k) * W + j] = t;
            }
        };
        // the next doubling (nuts.py:71-103 went on), signed step es: extend the end the new direction points to.  This wave's
        // state is the end of the direction just extended; the other end waits in T_OEND
        auto next_doubling = [&](double es) {
            const int nd = es < 0. ? -1 : 1;
            if (nd != dir) {
                double oq[4], op_[4], og[4];
                tv_ld(G::T_OEND + 0, oq); tv_ld(G::T_OEND + 1, op_); tv_ld(G::T_OEND + 2, og);
                tv_st(G::T_OEND + 0, q); tv_st(G::T_OEND + 1, p); tv_st(G::T_OEND + 2, g);
#pragma unroll
                for (int r = 0; r < 4; ++r) { q[r] = oq[r]; p[r] = op_[r]; g[r] = og[r]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) pdbl[r] = p[r];
            eps_t = es;
            dir = nd;
            i_leaf = 0;
            ann_eps = 0.;
            paused_end = false;
            imode = I_EVAL;
        };
        bf_sync();  // P0: the bookkeepers' first commands are posted
        int trip_no = -1;
        (void)trip_no;
        for (;;) {
            ++trip_no;
            STRACE(0, 0);
            if (FLG[0] == 0.) break;
            // ---- commands of the bookkeepers (posted before the barrier that ended the previous trip) ----
            {
                const int cmd = (int)XC[G::X_CMD * 16 + c];
                const int op = cmd & 7;
                if (op == SC_STOP) {
                    imode = I_IDLE;
                    paused_end = false;
                } else if (op == SC_NEW || op == SC_INIT) {
                    // an iteration starts at T_START: Tree.__init__ (nuts.py:24-43) -- both ends and p_sum are the start point
                    tv_ld(G::T_START + 0, q); tv_ld(G::T_START + 1, p); tv_ld(G::T_START + 2, g);
                    if (cmd & 8) { const double *sp = vxb; for (int r = 0; r < 4; ++r) var[r] = sp[64 * r]; }
                    tv_st(G::T_OEND + 0, q); tv_st(G::T_OEND + 1, p); tv_st(G::T_OEND + 2, g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { PSUM[r] = p[r]; pdbl[r] = p[r]; }
                    epoch_i = (int)XC[G::X_EPOCH * 16 + c];
                    eps_t = XC[G::X_EPS * 16 + c];
                    ann_eps = XC[G::X_ANN * 16 + c];
                    dir = eps_t < 0. ? -1 : 1;
                    depth = 0;
                    i_leaf = 0;
                    init_eval = op == SC_INIT;
                    paused_end = false;
                    imode = I_EVAL;
                } else if (op == SC_DBL) {
                    epoch_i = (int)XC[G::X_EPOCH * 16 + c];
                    depth = (int)XC[G::X_DEPTH * 16 + c];
                    next_doubling(XC[G::X_EPS * 16 + c]);
                } else if (op == SC_ANN) {
                    ann_eps = XC[G::X_ANN * 16 + c];
                }
                if (paused_end && ann_eps != 0.) {   // the announcement arrived after this chain reached the doubling's end
                    depth += 1;
                    epoch_i += 1;
                    next_doubling(ann_eps);
                }
            }
            // ---- phase A: first half of the leapfrog step (integration.py:80-85), B operands, the proof's partial ----
            const bool ev = imode != I_IDLE;
            const bool any_ev = bf_any(ev);   // (a trip in which every chain waits for its bookkeeper: barriers only)
            double xs[4];
            if (imode == I_EVAL) {
                const double dt = 0.5 * eps_t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = bf_fma(dt, g[r], p[r]);
                    q[r] = bf_fma(eps_t, var[r] * p[r], q[r]);
                }
            }
            if (any_ev) {
                double t_r2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xs[r] = ev ? q[r] : 0.;
                    XB[(0 * NS + 4 * j + r) * 64 + lane] = xs[r];
                    const double xm = xs[r] - c_mu[r];
                    XB[(1 * NS + 4 * j + r) * 64 + lane] = xm;
                    t_r2[r] = ev ? c_hd[r] * (xm * xm) : 0.;   // (the proof's weighted norm: bf_bound_lam_max_weighted)
                }
                double r2p = bf_xor32_add(bf_xor16_add(sum4(t_r2)));
                if (gq == 0) PB[j * 16 + c] = r2p;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[r] = 0.;
                if (gq == 0) PB[j * 16 + c] = 0.;
            }
            STRACE(0, 1);
            bf_sync();  // B1
            STRACE(0, 2);
            // ---- phase B: row tile j of S x and H (x - mu) ----
            double sx[4] = {0., 0., 0., 0.}, hv[4] = {0., 0., 0., 0.};
            const bool skipH = proof_holds();
            const bool extra = !skipH;
            if (any_ev) {
                bf_acc4 aS0 = bf_acc4_zero(), aS1 = bf_acc4_zero(), aH0 = bf_acc4_zero(), aH1 = bf_acc4_zero();
                constexpr int KS = W == 1 ? 1 : 2, KH = NS / KS;   // K halves of a matvec: the group kernel's association
#pragma unroll
                for (int s = 0; s < KH; ++s) {
                    aS0 = bf_mfma(afS[s], XB[(0 * NS + s) * 64 + lane], aS0);
                    if constexpr (KS == 2) aS1 = bf_mfma(afS[KH + s], XB[(0 * NS + KH + s) * 64 + lane], aS1);
                }
                n_trip += 1;
                n_trip_h += skipH ? 0 : 1;
                if (!skipH) {
#pragma unroll
                    for (int s = 0; s < KH; ++s) {
                        aH0 = bf_mfma(afH[s], XB[(1 * NS + s) * 64 + lane], aH0);
                        if constexpr (KS == 2) aH1 = bf_mfma(afH[KH + s], XB[(1 * NS + KH + s) * 64 + lane], aH1);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sx[r] = (KS == 2) ? aS0[r] + aS1[r] : aS0[r];
                    hv[r] = (KS == 2) ? aH0[r] + aH1[r] : aH0[r];
                }
            }
            STRACE(0, 3);
            // ---- phase C: gradient, second half step, the evaluation's partial sums ----
            double ge[4], pn[4];
            const double dt_c = (imode == I_IDLE) ? 0. : 0.5 * eps_t;
            bool fin = false, have_lp = false;
            double logp_new = 0.;
            double t_kin[4] = {0., 0., 0., 0.};
            if (any_ev) {
                double t_val[4], t_b2[4], t_a1[4], t_a2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double gn = sx[r] + c_lin[r];
                    t_val[r] = bf_fma(0.5 * xs[r], sx[r], c_lin[r] * xs[r]);
                    const double xm = xs[r] - c_mu[r];
                    t_b2[r] = xm * hv[r];
                    // outside the bound (bfhip_oob.h): a1 = (x - mu) . (S mu + lin), a2 = (x - mu) . S (x - mu)
                    t_a1[r] = xm * (c_smu[r] + c_lin[r]);
                    t_a2[r] = xm * (sx[r] - c_smu[r]);
                    ge[r] = gn;
                    pn[r] = bf_fma(dt_c, ge[r], p[r]);
                    t_kin[r] = pn[r] * (var[r] * pn[r]);
                }
                if (!extra) {
                    double e2[2] = {sum4(t_kin), sum4(t_val)};
                    post_n(G::E_KIN, e2);
                    fin = ev;  // inside the bound (proven): the value is complete (the bookkeepers add it up)
                } else {
                    double e5[5] = {sum4(t_kin), sum4(t_val), sum4(t_b2), sum4(t_a1), sum4(t_a2)};
                    post_n(G::E_KIN, e5);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) { ge[r] = 0.; pn[r] = 0.; }
            }
            if (__builtin_expect(extra, 0)) {   // (cold: spill code goes here, not into the proven trip)
                bf_sync();  // B2a: this trip's sums
                bool oob = false;
                if (ev) {
                    const double r_val = sumw(RBE + (G::E_VAL * W) * 16 + c, 16), r_b2 = sumw(RBE + (G::E_B2 * W) * 16 + c, 16);
                    double f = (m.c0 + r_val) + 0.;
                    const double a2 = m.alpha * m.alpha;
                    double bt = 0.;
                    if (!(r_b2 < a2 * (1. - 1e-12))) bt = bf_sqrt(r_b2);
                    if (bt > m.alpha) {
                        // outside the alpha-ellipsoid (poly.py:480-503): S x_0 of the projected point by linearity, no pass at x_0
                        // (bfhip_oob.h; the group kernel's expressions)
                        const double r_a1 = sumw(RBE + (G::E_A1 * W) * 16 + c, 16), r_a2 = sumw(RBE + (G::E_A2 * W) * 16 + c, 16);
                        const BfOob o = bf_oob_scalars(m.alpha, m.inv_alpha, m.f_mu, m.f_poly_mu, bt, r_a1, r_a2);
                        f = o.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ge[r] = bf_oob_grad(o, c_smu[r] + c_lin[r], sx[r] - c_smu[r], hv[r]);
                            pn[r] = bf_fma(dt_c, ge[r], p[r]);
                            t_kin[r] = pn[r] * (var[r] * pn[r]);
                        }
                        oob = true;
                    }
                    fin = true;
                    have_lp = true;
                    logp_new = f;
                }
                // the kinetic energy with the complete momentum (nobody reads E_KIN between B2a and B2; the chains inside the
                // bound post the number they posted before)
                if (bf_any(oob)) post(G::E_KIN, sum4(t_kin));
            }
            // ---- the finished leaf: its state, the U-turn sums of the subtrees it completes (nuts.py:146-161 per merge,
            // :88-101 per doubling), the momentum part of the subtree stack.  None of it depends on a random draw. ----
            const bool leafy = fin && !init_eval;
            const int leaf_epoch = epoch_i;   // (the epoch this leaf belongs to: an announced doubling starts the next one below)
            int nm = 0;
            if (fin) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { p[r] = pn[r]; g[r] = ge[r]; }  // integration.py:90
                tv_st(G::T_LEAF + 0, q); tv_st(G::T_LEAF + 1, g);
                imode = I_EVAL;
            }
            nm = leafy ? merge_levels(i_leaf, depth) : 0;
            {
                double tTL[4], tTPs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { tTL[r] = p[r]; tTPs[r] = p[r]; }
                if (bf_any(leafy && nm >= 1)) {  // level 0: the waiting leaf L0 and the new one (nuts.py:150-151)
                    double t0[4], t1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double ps0 = L0p[r] + p[r];
                        t0[r] = ps0 * (var[r] * L0p[r]);
                        t1[r] = ps0 * (var[r] * p[r]);
                        if (leafy && nm >= 1) { tTPs[r] = ps0; tTL[r] = L0p[r]; }
                    }
                    double m2[2] = {sum4(t0), sum4(t1)};
                    post_n(G::U_M0, m2);
                }
                for (int lev = 1; bf_any(leafy && lev < nm); ++lev) {
                    const bool on = leafy && lev < nm;
                    double A[4] = {0., 0., 0., 0.}, B[4] = {0., 0., 0., 0.}, S1[4] = {0., 0., 0., 0.};
                    if (on) { stk_ld(lev, 0, A); stk_ld(lev, 1, B); stk_ld(lev, 2, S1); }
                    double t[6][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double psum = S1[r] + tTPs[r];
                        const double vA = var[r] * A[r], vB = var[r] * B[r], vC = var[r] * tTL[r], vD = var[r] * p[r];
                        const double ps1 = S1[r] + tTL[r];   // :155-157
                        const double ps2 = B[r] + tTPs[r];   // :158-160
                        t[0][r] = psum * vA; t[1][r] = psum * vD; t[2][r] = ps1 * vA; t[3][r] = ps1 * vC;
                        t[4][r] = ps2 * vB; t[5][r] = ps2 * vD;
                        if (on) { tTL[r] = A[r]; tTPs[r] = psum; }
                    }
                    if (lev <= G::LSH) {
                        double s6[6];
#pragma unroll
                        for (int k = 0; k < 6; ++k) s6[k] = sum4(t[k]);
                        post_n(G::U_LV + 6 * (lev - 1), s6);
                    } else {
#pragma unroll
                        for (int k = 0; k < 6; ++k) post_lv(lev, k, sum4(t[k]));
                    }
                }
                const bool dbl_end = leafy && nm == depth;
                if (bf_any(dbl_end)) {  // the doubling completes: Tree.extend's checks, nuts.py:86-101
                    double Oe[4] = {0., 0., 0., 0.};
                    if (dbl_end) tv_ld(G::T_OEND + 1, Oe);
                    double t[6][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double Lp = dir > 0 ? Oe[r] : pdbl[r], Rp = dir > 0 ? pdbl[r] : Oe[r];   // the tree's ends before this doubling
                        const double ps = PSUM[r] + tTPs[r];  // :86 (in place)
                        const double vN = var[r] * p[r], vT = var[r] * tTL[r], vL = var[r] * Lp, vR = var[r] * Rp;
                        // (reference behaviour, kept on purpose: leftmost_p_sum / rightmost_p_sum alias the updated p_sum)
                        if (dir > 0) {
                            const double ps1 = ps + tTL[r], ps2 = Rp + tTPs[r];
                            t[0][r] = ps * vL; t[1][r] = ps * vN; t[2][r] = ps1 * vL; t[3][r] = ps1 * vT;
                            t[4][r] = ps2 * vR; t[5][r] = ps2 * vN;
                        } else {
                            const double ps1 = tTPs[r] + Lp, ps2 = tTL[r] + ps;
                            t[0][r] = ps * vN; t[1][r] = ps * vR; t[2][r] = ps1 * vN; t[3][r] = ps1 * vL;
                            t[4][r] = ps2 * vT; t[5][r] = ps2 * vR;
                        }
                        if (dbl_end) PSUM[r] = ps;
                    }
                    double s6[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) s6[k] = sum4(t[k]);
                    post_n(G::U_EXT, s6);
                }
                // the momentum part of what the tree does with this leaf (the bookkeepers do the rest when they get to it)
                if (leafy) {
                    if (dbl_end) {
                        // this state is the tree's new end; the next doubling goes where the bookkeepers announced, or waits
                        if (ann_eps != 0.) {
                            depth += 1;
                            epoch_i += 1;
                            next_doubling(ann_eps);
                        } else {
                            imode = I_IDLE;
                            paused_end = true;
                        }
                    } else {
                        if (nm == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) L0p[r] = p[r];
                        } else {
                            stk_st(nm, 0, tTL); stk_st(nm, 1, p); stk_st(nm, 2, tTPs);
                        }
                        i_leaf += 1;
                    }
                } else if (fin) {
                    imode = I_IDLE;       // the launch's first evaluation: the bookkeepers start the iteration
                }
            }
            if (wv == 0 && gq == 0) {
                XC[G::X_TAG * 16 + c] = fin ? (double)leaf_epoch : -1.;
                XC[G::X_HASLP * 16 + c] = have_lp ? 1. : 0.;
                XC[G::X_LOGP * 16 + c] = logp_new;
            }
            STRACE(0, 4);
            bf_sync();  // B2
            STRACE(0, 5);
        }
        if (a.gcount && tid == 0) {
            bf_atomic_add_u64(a.gcount, n_trip);
            bf_atomic_add_u64(a.gcount + 1, n_trip_h);
        }
        return;
    }

    // =========================================================================================================
    // BOOKKEEPER (the group kernel's tree logic, bfhip_group.h, on the leaf the integrators finished one trip ago)
    // =========================================================================================================
    const bool writer = j == 0 && gq == 0;
    const int nw = a.cfg.n_warmup;
    double var[4], L0q[4], L0g[4], PRq[4], PRg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        var[r] = 1.;
        L0q[r] = L0g[r] = PRq[r] = PRg[r] = 0.;
    }
    uint64_t rs[4] = {0, 0, 0, 0};
    int mode = M_DONE, i_iter = 0, err = 0, epoch = 0;
    bool ann_pending = false;   // a doubling was announced to the integrators under epoch + 1 and is not confirmed yet
    int dir = 1, depth = 0, i_leaf = 0, n_prop = 0, diverged = 0;
    double eps = 0., start_energy = 0., acc_sum = 0.;
    double max_de = 0., w_off = 0., tree_W = 1.;
    double L0_W = 0., L0_acc = 0., L0_E = 0., L0_logp = 0.;
    double prop_E = 0., prop_logp = 0.;
    // cold scalars in LDS.  All four bookkeeper waves need them at the end of an iteration and there is no barrier inside that
    // code, so the copy of parity (i_iter & 1) is only READ and the other one only WRITTEN (by every wave, with the same values)
    enum { K_LOG_STEP = 0, K_LOG_BAR, K_HBAR, K_SMU, K_COUNT, K_FG_N, K_BG_N, K_N_SAMPLES, K_PREV_UPD, K_ADAPT_WINDOW,
           K_STEP_NOW, K_STEP_BAR };   // (exp(log_step), exp(log_bar): what the tree steps with and the statistics report)
    static_assert(K_STEP_BAR + 1 == G::NCOLD, "cold scalars");
    auto cold = [&](int it) -> double * { return COLD + (it & 1) * (G::NCOLD * 16) + c; };
    bool need_E0 = false;
    unsigned long long nlf = 0;

    double *scp = a.sc + (size_t)(real ? chain : 0) * BFHIP_SC_N;
    double *vecp = a.vec + (size_t)(real ? chain : 0) * BFHIP_VEC_N * d;
    double *lsc = LS + c * LSS;

    auto load_vec = [&](int field, double (&v)[4], double pad) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dim = dbase + 4 * r;
            v[r] = (dim < d) ? vecp[field * d + dim] : pad;
        }
    };
    auto store_vec = [&](int field, const double (&v)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dim = dbase + 4 * r;
            if (dim < d) vecp[field * d + dim] = v[r];
        }
    };
    auto rd_n = [&](int vi0, auto &out) {
        constexpr int N = sizeof(out) / sizeof(double);
        double t[N][W];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) t[i][w2] = RBE[((vi0 + i) * W + w2) * 16 + c];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if constexpr (W == 4) out[i] = (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
            else if constexpr (W == 2) out[i] = t[i][0] + t[i][1];
            else out[i] = t[i][0];
        }
    };
    auto rd_lv = [&](int lev, int k) -> double {
        if (lev <= G::LSH) return sumw(RBE + ((G::U_LV + 6 * (lev - 1) + k) * W) * 16 + c, 16);
        return sumw(rbg + (6 * (lev - 1 - G::LSH) + k) * W, 1);
    };
    // metric.random (metrics.py:83-86), the group kernel's stream: one xoshiro draw keys a SplitMix64 counter stream,
    // Box-Muller pairs shared between the rows g and g ^ 1
    auto draw_momentum = [&](bool on, double (&pnew)[4]) {
        uint64_t K = 0;
        if (on) K = bf_xoshiro_next(rs);
        const int godd = gq & 1;
        double mine[2], theirs[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 2 * godd + h;
            const uint64_t P = (uint64_t)((dbase + 4 * r) >> 1);
            const double u1 = bf_u01_open0(bf_mix64(K + (2 * P + 1) * BF_GOLDEN));
            const double u2 = bf_u01(bf_mix64(K + (2 * P + 2) * BF_GOLDEN));
            const double rad = bf_sqrt(-2. * bf_log(u1));
            double sn, cs;
            bf_sincospi(2. * u2, &sn, &cs);
            mine[h] = godd ? rad * sn : rad * cs;
            theirs[h] = godd ? rad * cs : rad * sn;
        }
        theirs[0] = bf_xor16_get(theirs[0]);
        theirs[1] = bf_xor16_get(theirs[1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double z = ((r >> 1) == godd) ? mine[r & 1] : theirs[r & 1];
            pnew[r] = (on && dbase + 4 * r < d) ? (1. / bf_sqrt(var[r])) * z : 0.;  // var^-1/2 z, metrics.py:83-86
        }
    };
    // command to the integrators (a new epoch): cmd, the signed step, the depth of the tree
    auto command = [&](int cmd, double eps_s, int dep, double ann = 0.) {
        epoch += ann_pending ? 2 : 1;   // (past an announced epoch: what the integrators computed under it is dropped)
        ann_pending = ann != 0.;
        if (writer) {
            XC[G::X_CMD * 16 + c] = (double)cmd;
            XC[G::X_EPOCH * 16 + c] = (double)epoch;
            XC[G::X_EPS * 16 + c] = eps_s;
            XC[G::X_DEPTH * 16 + c] = (double)dep;
            XC[G::X_ANN * 16 + c] = ann;
        }
    };
    // The direction of the doubling after the one that ends with the next leaf, read ahead: processing that leaf will draw once
    // per merge level (`dep` of them), once for the proposal swap (nuts.py:81-83), then the direction (:210).  0: no next doubling.
    auto look_ahead = [&](int dep) -> int {
        if (dep + 1 >= a.cfg.max_treedepth) return 0;
        uint64_t r2[4] = {rs[0], rs[1], rs[2], rs[3]};
        for (int k = 0; k < dep + 1; ++k) (void)bf_xoshiro_next(r2);
        return (bf_u01(bf_xoshiro_next(r2)) < 0.5) ? 1 : -1;
    };
    // Tree.__init__ (nuts.py:24-43) at (sq, pnew, sg): the proposal is the starting point; the integrators start there
    auto tree_reset = [&](const double (&sq)[4], const double (&pnew)[4], const double (&sg)[4], bool reload_var) {
        tree_W = 1.;
        w_off = 0.;
        max_de = 0.;
        depth = 0; acc_sum = 0.; n_prop = 0; diverged = 0; i_leaf = 0;
        eps = cold(i_iter)[((i_iter < nw) ? K_STEP_NOW : K_STEP_BAR) * 16];  // step_size.py:25-29: exp(log_step) / exp(log_bar)
        dir = (bf_u01(bf_xoshiro_next(rs)) < 0.5) ? 1 : -1;  // nuts.py:210, log(U) < log(1/2)
        tv_st(G::T_START + 0, sq); tv_st(G::T_START + 1, pnew); tv_st(G::T_START + 2, sg);
#pragma unroll
        for (int r = 0; r < 4; ++r) { PRq[r] = sq[r]; PRg[r] = sg[r]; }
        if (reload_var) { double *sp = vxb; for (int r = 0; r < 4; ++r) sp[64 * r] = var[r]; }
        // (the first doubling is one leaf: the direction of the second travels with this command)
        command(SC_NEW | (reload_var ? 8 : 0), eps * (double)dir, 0, eps * (double)look_ahead(0));
        mode = M_LEAF;
        // the start energy needs the kinetic energy of the new momentum: its partial sums cross the next barrier
        double t_k0[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) t_k0[r] = pnew[r] * (var[r] * pnew[r]);  // metrics.py:88-91
        const double kp = sum4(t_k0);
        need_E0 = true;
        return kp;
    };

    double q0[4];
    if (real) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rs[k] = a.rng[(size_t)chain * 4 + k];
        i_iter = (int)scp[BFHIP_SC_I_ITER];
        {
            double *cw = cold(i_iter);
            cw[K_LOG_STEP * 16] = scp[BFHIP_SC_LOG_STEP];
            cw[K_LOG_BAR * 16] = scp[BFHIP_SC_LOG_BAR];
            cw[K_HBAR * 16] = scp[BFHIP_SC_HBAR];
            cw[K_SMU * 16] = scp[BFHIP_SC_MU];
            cw[K_COUNT * 16] = scp[BFHIP_SC_COUNT];
            cw[K_FG_N * 16] = scp[BFHIP_SC_FG_N];
            cw[K_BG_N * 16] = scp[BFHIP_SC_BG_N];
            cw[K_N_SAMPLES * 16] = scp[BFHIP_SC_N_SAMPLES];
            cw[K_PREV_UPD * 16] = scp[BFHIP_SC_PREV_UPDATE];
            cw[K_ADAPT_WINDOW * 16] = scp[BFHIP_SC_ADAPT_WINDOW];
            cw[K_STEP_NOW * 16] = bf_exp(scp[BFHIP_SC_LOG_STEP]);
            cw[K_STEP_BAR * 16] = bf_exp(scp[BFHIP_SC_LOG_BAR]);
        }
        err = (int)scp[BFHIP_SC_ERROR];
        load_vec(BFHIP_VEC_Q, q0, 0.);
        load_vec(BFHIP_VEC_VAR, var, 1.);
#pragma unroll
        for (int r = 0; r < 4; ++r) PRq[r] = q0[r];
        if (i_iter < a.iter_end && err == 0) mode = M_INIT;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) q0[r] = 0.;
    }
    {
        // the launch's first iteration evaluates its starting point: a step of length 0 from (q0, 0, 0) (base_hmc.py:70)
        double z4[4] = {0., 0., 0., 0.};
        if (mode == M_INIT) {
            tv_st(G::T_START + 0, q0); tv_st(G::T_START + 1, z4); tv_st(G::T_START + 2, z4);
            double *sp = vxb;
            for (int r = 0; r < 4; ++r) sp[64 * r] = var[r];
            command(SC_INIT | 8, 0., 0);
        } else if (writer) {
            XC[G::X_CMD * 16 + c] = (double)SC_STOP;
            XC[G::X_EPOCH * 16 + c] = 0.;
            XC[G::X_EPS * 16 + c] = 0.;
            XC[G::X_DEPTH * 16 + c] = 0.;
            XC[G::X_ANN * 16 + c] = 0.;
        }
        if (writer) XC[G::X_TAG * 16 + c] = -1.;
        const bool alive = bf_any(mode != M_DONE);
        if (tid == W * 64) FLG[0] = alive ? 1. : 0.;
    }
    bf_sync();  // P0

    int trip_no = -1;
    (void)trip_no;
    for (;;) {
        ++trip_no;
        STRACE(1, 0);
        if (FLG[0] == 0.) break;
        // ---- the leaf the integrators finished in the previous trip (if it belongs to this epoch) and ALL its sums: nothing
        // of this trip is needed, so everything is read before the first barrier ----
        const bool have = mode != M_DONE && (int)XC[G::X_TAG * 16 + c] == epoch;
        const bool is_leaf = have && mode == M_LEAF;
        const int nm = is_leaf ? merge_levels(i_leaf, depth) : 0;
        const bool any_m0 = bf_any(is_leaf && nm >= 1);
        const bool any_lv1 = bf_any(is_leaf && nm >= 2);
        const bool any_ext = bf_any(is_leaf && nm == depth);
        const bool any_e0 = bf_any(need_E0);
        double lq[4], lg[4];
        tv_ld(G::T_LEAF + 0, lq); tv_ld(G::T_LEAF + 1, lg);
        double sv_e[2], sv_m0[2] = {1., 1.}, sv_l1[6] = {1., 1., 1., 1., 1., 1.}, sv_x[6] = {1., 1., 1., 1., 1., 1.};
        rd_n(G::E_KIN, sv_e);
        if (any_m0) rd_n(G::U_M0, sv_m0);
        if (any_lv1) rd_n(G::U_LV, sv_l1);
        if (any_ext) rd_n(G::U_EXT, sv_x);
        bool turn_deep = false;   // U-turn checks of the merge levels above 1, in the order the merges take them
        int lev_turn = 1 << 30;   // the first level >= 2 whose check says turning
        if (bf_any(is_leaf && nm >= 3)) {
            for (int lev = 2; lev < nm; ++lev) {
                bool t = false;
#pragma unroll
                for (int k = 0; k < 6; ++k) t = t || (rd_lv(lev, k) <= 0.);
                if (t && !turn_deep) { turn_deep = true; lev_turn = lev; }
            }
        }
        const double logp_new = (XC[G::X_HASLP * 16 + c] != 0.) ? XC[G::X_LOGP * 16 + c] : (m.c0 + sv_e[1]) + 0.;
        const double kin = sv_e[0];
        double kin0 = 0.;
        if (any_e0) kin0 = sumw(RBK + c, 16);
        STRACE(1, 1);
        bf_sync();  // B1
        STRACE(1, 2);
        const bool extra = !proof_holds();
        if (writer) XC[G::X_CMD * 16 + c] = (double)SC_CONT;   // (the integrators took the previous trip's command before B1)

        const double E_new = 0.5 * kin - logp_new;  // integration.py:92-93
        // the start energy of an iteration that began at the end of the previous trip
        if (need_E0 && mode != M_DONE) {
            const double E0 = 0.5 * kin0 - prop_logp;  // integration.py:28-34
            if (!(bf_fabs(E0) <= BF_DBL_MAX)) err = 1;           // base_hmc.py:72-76
            start_energy = E0;
            prop_E = E0;
            need_E0 = false;
        }
        STRACE(1, 3);
        // ---- per-chain state machine (bfhip_group.h) ----
        enum { S_NONE, S_MERGE, S_ABORT, S_DBL_END, S_END };
        int st = S_NONE, lev = 0, src = -1;
        double dE = 0., aw = 0.;
        double T_W = 0., T_acc = 0., T_E = 0., T_logp = 0.;   // the finished subtree: weight, accept sum, its proposal's energy and logp
        bool resc = false;
        const bool ok = have && err == 0;
        const bool is_init = ok && mode == M_INIT, leaf = ok && mode == M_LEAF;
        double kin0_part = 0.;
        bool post_k0 = false;
        if (bf_any(is_init)) {
            if (is_init) {
                // BaseHMC.astep start (base_hmc.py:70-76): the value and gradient at the chain's position; the start energy
                // follows with the kinetic energy of the momentum drawn now
                prop_logp = logp_new;
                if (!(bf_fabs(logp_new) <= BF_DBL_MAX)) err = 1;
            }
            double pnew[4];
            draw_momentum(is_init, pnew);   // metric.random (the first draw of the chain's stream in this launch)
            if (is_init && err == 0) { kin0_part = tree_reset(lq, pnew, lg, false); post_k0 = true; }
        }
        if (leaf) {
            nlf += 1;
            n_prop += 1;
            dE = E_new - start_energy;
            dE = (dE != dE) ? __builtin_inf() : dE;
            max_de = (bf_fabs(dE) > bf_fabs(max_de)) ? dE : max_de;
            T_E = E_new;
            T_logp = logp_new;
            T_acc = 0.;
            const bool dv = !(bf_fabs(dE) < a.cfg.max_change);
            diverged = dv ? 1 : diverged;
            st = dv ? S_ABORT : S_MERGE;
            aw = dv ? 0. : -dE - w_off;
            resc = aw > 600.;
        }
        if (resc) {  // (rare) the weights follow a new offset; the stacked subtrees' weights are rescaled when they are read
            const double sc_ = bf_exp(-aw);
            tree_W = tree_W * sc_;
            L0_W *= sc_;
            w_off = w_off + aw;
            aw = 0.;
        }
        if (st == S_MERGE) {
            T_W = bf_exp(aw);
            const double pacc = (w_off == 0.) ? T_W : bf_exp(-dE);
            T_acc = pacc > 1. ? 1. : pacc;
            if (nm >= 1) {
                // ---- level-0 merge with the waiting leaf L0 (nuts.py:146-178) ----
                const double d0 = sv_m0[0], d1 = sv_m0[1];
                T_acc = L0_acc + T_acc;  // :173
                const double Wsum = L0_W + T_W;
                if (Wsum != Wsum) err = 2;
                const double u = bf_u01(bf_xoshiro_next(rs));  // :163-167, drawn even when turning
                lev = 1;
                if ((d0 <= 0.) || (d1 <= 0.)) {
                    st = S_ABORT;
                } else {
                    if (!((u * Wsum < T_W) || (u == 0.))) { src = 0; T_E = L0_E; T_logp = L0_logp; }
                    T_W = Wsum;
                }
            }
            while (st == S_MERGE && lev < nm) {
                bool turning = false;
                if (lev == 1) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) turning = turning || (sv_l1[k] <= 0.);
                } else {
                    turning = lev == lev_turn;   // (read before the barrier; levels above a turning one are never reached)
                }
                const double *lsp = lsc + lev * SS_N;
                double lw = lsp[SS_W];
                if (lsp[SS_OFF] != w_off) lw *= bf_exp(lsp[SS_OFF] - w_off);   // stored under an older offset
                T_acc = lsp[SS_ACC] + T_acc;  // :173
                const double Wsum = lw + T_W;
                if (Wsum != Wsum) err = 2;
                const double u = bf_u01(bf_xoshiro_next(rs));  // consumed even when this merge's check says turning
                const bool keep_t2 = (u * Wsum < T_W) || (u == 0.);
                if (turning) {
                    st = S_ABORT;
                } else {
                    if (!keep_t2) { src = lev; T_E = lsp[SS_E]; T_logp = lsp[SS_LOGP]; }
                    T_W = Wsum;
                }
                lev += 1;
            }
            if (st == S_MERGE) {
                if (lev < depth) {
                    // the subtree waits for its right sibling: the proposal part (the integrators keep the momenta)
                    if (lev == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { L0q[r] = lq[r]; L0g[r] = lg[r]; }
                        L0_W = T_W; L0_acc = T_acc; L0_E = E_new; L0_logp = logp_new;
                    } else {
                        double tq[4], tg[4];
                        if (src < 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { tq[r] = lq[r]; tg[r] = lg[r]; }
                        } else if (src == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { tq[r] = L0q[r]; tg[r] = L0g[r]; }
                        } else {
                            stk_ld(src, 3, tq);
                            stk_ld(src, 4, tg);
                        }
                        stk_st(lev, 3, tq); stk_st(lev, 4, tg);
                        if (writer) {
                            double *lsp = lsc + lev * SS_N;
                            lsp[SS_W] = T_W; lsp[SS_ACC] = T_acc; lsp[SS_E] = T_E; lsp[SS_LOGP] = T_logp; lsp[SS_OFF] = w_off;
                        }
                    }
                    i_leaf += 1;
                    st = S_NONE;
                    if (merge_levels(i_leaf, depth) == depth) {   // the next leaf completes this doubling: announce the one after
                        const int nd = look_ahead(depth);
                        if (nd != 0) {
                            ann_pending = true;
                            if (writer) { XC[G::X_CMD * 16 + c] = (double)SC_ANN; XC[G::X_ANN * 16 + c] = eps * (double)nd; }
                        }
                    }
                } else {
                    st = S_DBL_END;
                }
            }
        }
        const bool any_end = bf_any(st != S_NONE);
        STRACE(1, 6);
        if (any_end) {
            if (st == S_ABORT) {
                // unwind: every pending ancestor adds its left half's accept_sum (nuts.py:173)
                for (int al = (diverged ? 0 : lev); al < depth; ++al)
                    if ((i_leaf >> al) & 1) T_acc = (al == 0 ? L0_acc : lsc[al * SS_N + SS_ACC]) + T_acc;
                depth += 1;  // nuts.py:71-73
                acc_sum += T_acc;
                st = S_END;
            } else if (st == S_DBL_END) {
                // ---- Tree.extend after a complete subtree: nuts.py:71-103 ----
                depth += 1;
                acc_sum += T_acc;
                {   // :81-83  logbern(ls_new - ls_old)  <=>  U * W_old < W_new
                    if (T_W != T_W || tree_W != tree_W) err = 2;
                    const double u = bf_u01(bf_xoshiro_next(rs));
                    if ((u * tree_W < T_W) || (u == 0.)) {
                        if (src < 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { PRq[r] = lq[r]; PRg[r] = lg[r]; }
                        } else if (src == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { PRq[r] = L0q[r]; PRg[r] = L0g[r]; }
                        } else {
                            stk_ld(src, 3, PRq);
                            stk_ld(src, 4, PRg);
                        }
                        prop_E = T_E;
                        prop_logp = T_logp;
                    }
                    tree_W = tree_W + T_W;  // :85
                }
                bool turning = false;
#pragma unroll
                for (int k = 0; k < 6; ++k) turning = turning || (sv_x[k] <= 0.);
                if (turning || depth >= a.cfg.max_treedepth) {
                    st = S_END;
                } else {
                    const int nd = (bf_u01(bf_xoshiro_next(rs)) < 0.5) ? 1 : -1;  // nuts.py:210
                    dir = nd;
                    i_leaf = 0;
                    if (ann_pending) {      // announced (the same draw, read ahead): the integrators are already there
                        epoch += 1;
                        ann_pending = false;
                    } else {
                        command(SC_DBL, eps * (double)nd, depth);   // the integrators extend that end
                    }
                    st = S_NONE;
                }
            }
            // ================= iteration end: base_hmc.py:80-85 =================
            bool new_iter = false, reload = false;
            if (st == S_END && err == 0) {
                const bool warm = i_iter < nw;
                const double accept_stat = acc_sum / (double)n_prop;  // nuts.py:186
                const double *cr = cold(i_iter);
                double log_step = cr[K_LOG_STEP * 16], log_bar = cr[K_LOG_BAR * 16], hbar = cr[K_HBAR * 16], count = cr[K_COUNT * 16];
                const double smu = cr[K_SMU * 16];
                double fg_n = cr[K_FG_N * 16], bg_n = cr[K_BG_N * 16], n_samples = cr[K_N_SAMPLES * 16], prev_upd = cr[K_PREV_UPD * 16];
                double adapt_window = cr[K_ADAPT_WINDOW * 16];
                double step_now = cr[K_STEP_NOW * 16], step_bar = cr[K_STEP_BAR * 16];
                if (warm && a.cfg.adapt_step_size) {  // step_size.py:31-45
                    const double wgt = 1. / (count + a.cfg.t_0);
                    hbar = ((1. - wgt) * hbar + wgt * (a.cfg.target_accept - accept_stat));
                    log_step = smu - hbar * bf_sqrt(count) / a.cfg.gamma;
                    const double mk = bf_exp(-a.cfg.k * bf_log(count));  // count ** -k
                    log_bar = mk * log_step + (1. - mk) * log_bar;
                    count = count + 1.;
                    step_now = bf_exp(log_step);
                    step_bar = bf_exp(log_bar);
                }
                // the proposal (PRq, PRg) is the new sample and the start of the next iteration
                const int orow = i_iter - a.iter_out0;
                if (orow >= 0 && orow < a.n_out) {
                    if (writer) {
                        double *sp = a.stats + ((size_t)chain * a.n_out + orow) * BFHIP_STAT_STRIDE;
                        sp[BFHIP_NS_LOGP] = prop_logp;
                        sp[BFHIP_NS_ENERGY] = prop_E;
                        sp[BFHIP_NS_TREE_DEPTH] = (double)depth;
                        sp[BFHIP_NS_TREE_SIZE] = (double)n_prop;
                        sp[BFHIP_NS_MEAN_TREE_ACCEPT] = accept_stat;
                        sp[BFHIP_NS_STEP_SIZE] = step_now;
                        sp[BFHIP_NS_STEP_SIZE_BAR] = step_bar;
                        sp[BFHIP_NS_WARMUP] = warm ? 1. : 0.;
                        sp[BFHIP_NS_ENERGY_CHANGE] = prop_E - start_energy;
                        sp[BFHIP_NS_MAX_ENERGY_CHANGE] = max_de;
                        sp[BFHIP_NS_DIVERGING] = (double)diverged;
                    }
                    double *xp = a.samples + ((size_t)chain * a.n_out + orow) * d;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (dbase + 4 * r < d) xp[dbase + 4 * r] = PRq[r];
                }
                bool var_changed = false;
                // QuadMetricDiagAdapt.update: metrics.py:186-211, _WeightedVariance.add_sample :354-360
                if (warm && a.cfg.adapt_metric) {
                    const long delta = (long)(n_samples - prev_upd);
                    double fm[4], fr[4], bm[4], br[4];
                    load_vec(BFHIP_VEC_FG_MEAN, fm, 0.);
                    load_vec(BFHIP_VEC_FG_RAW, fr, 0.);
                    load_vec(BFHIP_VEC_BG_MEAN, bm, 0.);
                    load_vec(BFHIP_VEC_BG_RAW, br, 0.);
                    fg_n += 1.;
                    bg_n += 1.;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double od = PRq[r] - fm[r];
                        fm[r] += od / fg_n;
                        fr[r] += 1. * od * (PRq[r] - fm[r]);
                        od = PRq[r] - bm[r];
                        bm[r] += od / bg_n;
                        br[r] += 1. * od * (PRq[r] - bm[r]);
                    }
                    if ((delta + 1) % (long)a.cfg.update_window == 0) {  // metrics.py:181-184
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (dbase + 4 * r < d) var[r] = fr[r] / fg_n;
                        store_vec(BFHIP_VEC_VAR, var);
                        var_changed = true;
                    }
                    if ((double)delta >= adapt_window) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { fm[r] = bm[r]; fr[r] = br[r]; bm[r] = 0.; br[r] = 0.; }
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
                }
                i_iter += 1;
                {
                    double *cw = cold(i_iter);
                    cw[K_LOG_STEP * 16] = log_step; cw[K_LOG_BAR * 16] = log_bar; cw[K_HBAR * 16] = hbar; cw[K_SMU * 16] = smu;
                    cw[K_COUNT * 16] = count; cw[K_FG_N * 16] = fg_n; cw[K_BG_N * 16] = bg_n; cw[K_N_SAMPLES * 16] = n_samples;
                    cw[K_PREV_UPD * 16] = prev_upd; cw[K_ADAPT_WINDOW * 16] = adapt_window;
                    cw[K_STEP_NOW * 16] = step_now; cw[K_STEP_BAR * 16] = step_bar;
                }
                if (i_iter < a.iter_end) {
                    new_iter = true;
                    reload = var_changed;   // (travels with the new iteration's command)
                } else {
                    mode = M_DONE;
                    command(SC_STOP, 0., 0);
                }
            }
            STRACE(1, 7);
            // next iteration: metric.random, then the tree starts at the proposal with its value and gradient
            if (bf_any(new_iter)) {
                double pnew[4];
                draw_momentum(new_iter, pnew);
                if (new_iter) {
                    double sq[4], sg[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sq[r] = PRq[r]; sg[r] = PRg[r]; }
                    kin0_part = tree_reset(sq, pnew, sg, reload);
                    post_k0 = true;
                }
            }
        }
        if (bf_any(post_k0)) {   // the new momentum's kinetic energy, summed over the bookkeeper waves across the next barrier
            const double t = bf_xor32_add(bf_xor16_add(post_k0 ? kin0_part : 0.));
            if (gq == 0 && post_k0) RBK[j * 16 + c] = t;
        }
        if (err != 0 && mode != M_DONE) {
            mode = M_DONE;
            command(SC_STOP, 0., 0);
        }
        STRACE(1, 4);
        if (extra) bf_sync();  // B2a (the integrators read their sums)
        {
            const bool alive = bf_any(mode != M_DONE);
            if (tid == W * 64) FLG[0] = alive ? 1. : 0.;
        }
        bf_sync();  // B2
        STRACE(1, 5);
    }

    // ---- write the chain state back ----
    if (real) {
        store_vec(BFHIP_VEC_Q, PRq);
        if (writer) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.rng[(size_t)chain * 4 + k] = rs[k];
            const double *cr = cold(i_iter);
            scp[BFHIP_SC_LOG_STEP] = cr[K_LOG_STEP * 16];
            scp[BFHIP_SC_LOG_BAR] = cr[K_LOG_BAR * 16];
            scp[BFHIP_SC_HBAR] = cr[K_HBAR * 16];
            scp[BFHIP_SC_COUNT] = cr[K_COUNT * 16];
            scp[BFHIP_SC_FG_N] = cr[K_FG_N * 16];
            scp[BFHIP_SC_BG_N] = cr[K_BG_N * 16];
            scp[BFHIP_SC_N_SAMPLES] = cr[K_N_SAMPLES * 16];
            scp[BFHIP_SC_PREV_UPDATE] = cr[K_PREV_UPD * 16];
            scp[BFHIP_SC_ADAPT_WINDOW] = cr[K_ADAPT_WINDOW * 16];
            scp[BFHIP_SC_I_ITER] = (double)i_iter;
            scp[BFHIP_SC_ERROR] = (double)err;
            if (a.n_leapfrog && nlf) bf_atomic_add_u64(a.n_leapfrog, nlf);
        }
    }
}
