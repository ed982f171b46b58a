// bfhip_group.h -- NUTS / HMC transitions for the common surrogate at d <= 64, "group" layout (gfx950).
//
// Replaces, for all chains of a launch: BaseHMC.run / astep (samplers/hmc_utils/base_hmc.py:62-85,155-156), the NUTS
// tree (samplers/nuts.py:21-217) or the static HMC trajectory (samplers/hmc.py:16-49), CpuLeapfrogIntegrator._step
// (samplers/hmc_utils/integration.py:68-95), Density.logp_and_grad on the linear + quadratic surrogate with its
// extrapolation bound (core/density.py:724-754, modules/poly.py:466-503), DualAverageAdaptation
// (samplers/hmc_utils/step_size.py:10-51) and QuadMetricDiagAdapt (samplers/hmc_utils/metrics.py:135-237,333-371).
//
// Layout.  A workgroup of W = DP/16 waves owns a GROUP of 16 chains for the whole launch.  Lane l = (c = l & 15,
// g = l >> 4) of wave j holds, for chain c, the four dimensions 16 j + g + 4 r (r = 0..3) of every state vector:
//   * Per-chain scalars (tree weights, accept sums, step-size state, xoshiro state, counters) live ONCE PER LANE: one
//     instruction advances the scalar logic of all 16 chains (the sliced kernel bf_sampler_kernel spends one wave per
//     chain, i.e. one instruction per chain, on it).  Control flow is per lane (exec masks), not per wave.
//   * The gradient tiles need no transposition: wave j computes row tile j of S X^T and H (X - mu)^T with
//     v_mfma_f64_16x16x4_f64, whose result registers are exactly the wave's own elements (D[4 r + g][c]); the B operand
//     of k-step s is element s & 3 of wave s >> 2, exchanged through LDS (XB) once per trip.
//   * Dot products over the dimensions (energies, the value of the surrogate, the bound test, the U-turn checks) are
//     summed per lane over r, over g by two row swaps, and over the waves through LDS (RB): ONE exchange per trip carries
//     the evaluation's sums AND every U-turn check the finished leaf can trigger -- which subtrees merge after leaf i of
//     a doubling is known beforehand (the trailing one bits of i), and the vectors involved do not depend on the
//     multinomial draws -- so a trip is: first half step + operands | barrier | 4 W .. 8 W MFMAs + partial sums |
//     barrier | scalar logic + vector bookkeeping.  Two workgroup barriers per leapfrog step, no speculation, no wasted
//     evaluation;
//     an iteration starts from its predecessor's proposal, whose value and gradient travel with it through the merges
//     (base_hmc.py:70 evaluates it again; the numbers are the same).
//   * Summation order is fixed and the same as the sliced kernel's butterfly: r (dims 4 apart), then g, then waves; the
//     K halves of a matvec are added as part0 + part1.  A chain's results do not depend on its group or lane.  Two
//     (four) sums are reduced over g by one row swap per step (bf_pair16_add / bf_pair32_add: the swap of two different
//     registers is a transposed reduction) -- the same additions, value k ending up in row k.
//   * The extrapolation bound's test is decided without its H (x - mu) tiles whenever lam_max(H) |x - mu|^2 < alpha^2
//     proves every chain of the group inside (the outcome of the full test, so results do not depend on it); likewise
//     the decay term.  The common trip then skips its rare branches as one (wave-uniform guards): per-lane branches
//     cost a round trip through the scalar unit each, and with one wave per SIMD nothing hides it.
//
// The subtree stack (left p, right p, p_sum, proposal q, proposal gradient per level) keeps level 0 in registers, level 1
// in LDS next to the tree's ends, proposal and p_sum, and deeper levels in the context's global scratch.
//
// FS (feature set): 1 = linear + quadratic configs with the bound; bit 1 (2) = decay penalty (density.py:740-746);
// bit 2 (4) = constraint transform (density.py:92-140,747-750); bit 3 (8, round 6) = the PIPELINE DENSITY (bfhip_pld.h: multi-output
// surrogate + Gaussian likelihood + prior) instead of the single-output polynomial: the wave-per-chain kernel spends a
// wave per chain on the tree logic between two evaluations and 28 k cycles on a trip of EIGHT chains at the DES shape
// (profiles/r06b_trace_pld_des.log); here the 16 chains of a group share every instruction of it.  The evaluation is always "late"
// (the gradient needs the contractions' result): phase A | B1 | the bound's tiles, sums | B2 | evaluation points | P0 | monomials |
// P1 | F = C' Phi | P2 | W = C'^T r | P3 | gradient gather, second half step, U-turn sums | B3 | the state machine as before.
#pragma once
#include "bfhip_lane.h"
#include "bfhip_sampler_defs.h"
#include "bfhip_oob.h"
#include "bfhip_pld.h"

#define BF_DBL_MAX 1.7976931348623157e308

// tuning builds (tools/gvariant.sh -DBF_GTRACE=<n>): lane 0 of workgroup 0 stamps the cycle counter at up to 8 points
// of each of its first n trips into a.stamps
#if defined(BF_GTRACE2) && !defined(BF_HOST_EMU)
// second set of stamp points (inside the state machine): -DBF_GTRACE2=<n>
#define GTRACE(k) do { if ((k) == 0 && tid == 0 && bf_group() == 0 && a.stamps && trip_no < BF_GTRACE2) a.stamps[trip_no * 8] = __builtin_readcyclecounter(); } while (0)
#define GTRACE2(k) do { if (tid == 0 && bf_group() == 0 && a.stamps && trip_no < BF_GTRACE2) a.stamps[trip_no * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#elif defined(BF_GTRACE) && !defined(BF_HOST_EMU)
#define GTRACE2(k) do { } while (0)
#define GTRACE(k) do { if (tid == 0 && bf_group() == 0 && a.stamps && trip_no < BF_GTRACE) a.stamps[trip_no * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define GTRACE(k) do { } while (0)
#define GTRACE2(k) do { } while (0)
#endif

// tuning builds of the pipeline form (-DBF_PTRACE=<n>): 16 stamp points per trip (tools/trace_group_pld.py)
#if defined(BF_PTRACE) && !defined(BF_HOST_EMU)
#define PTRACE(k) do { if (tid == 0 && bf_group() == 0 && a.stamps && trip_no < BF_PTRACE) a.stamps[trip_no * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PTRACE(k) do { } while (0)
#endif

// knock-out builds (tools/gvariant.sh -DBF_KO_...): WRONG results on purpose, to bound what a piece of the trip costs on the
// critical path (the tree shapes do not depend on the pieces knocked out); never part of the library
#if defined(BF_KO_EXP) && !defined(BF_HOST_EMU)
#define BF_LEAF_EXP(x) (1. + 1e-3 * (x))
#else
#define BF_LEAF_EXP(x) bf_exp(x)
#endif
#if defined(BF_KO_RNG) && !defined(BF_HOST_EMU)
#define BF_TREE_DRAW(rs) ((rs)[0] += 0x9E3779B97F4A7C15ULL, (rs)[0])
#else
#define BF_TREE_DRAW(rs) bf_xoshiro_next(rs)
#endif

template <int W>
struct GroupGeo {
    static constexpr int DP = 16 * W, NS = 4 * W;
    static constexpr int KS = (W == 2 || W == 4) ? 2 : 1;  // K halves of a matvec (same association as the sliced kernel)
    static constexpr int MAXL = BFHIP_MAX_TREEDEPTH;
    static constexpr int LSS = 4 * MAXL + 1;                 // chain stride of the stack scalars
    // tree vectors in LDS, [slot][dimension][chain] (a wave's access covers 64 consecutive doubles): stack level 1
    // (left p, right p, p_sum, proposal q, proposal gradient), both ends of the tree (q, p, grad), its proposal
    // (q, grad) and its p_sum.  Deeper stack levels live in the context's global scratch.
    static constexpr int T_STK1 = 0, T_LEFT = 5, T_RIGHT = 8, T_PROP = 11, T_PSUM = 13, NTV = 14;
    // exchanged sums: evaluation, level-0 merge, merge levels 1..LSH, the doubling's checks in LDS; the merge levels
    // above LSH (one leaf in 2^LSH reaches them) go through global scratch
    static constexpr int LSH = 3;
    static constexpr int V_KIN = 0, V_VAL = 1, V_B2 = 2, V_BD2 = 3, V_LOGDET = 4, V_KIN0 = 5, V_M0 = 6;
    static constexpr int V_LV = V_M0 + 2, V_EXT = V_LV + 6 * LSH, NVAL = V_EXT + 6;
    static constexpr int NDEEP = 6 * (MAXL - 1 - LSH);       // sums of the merge levels LSH+1 .. MAXL-1
    static constexpr size_t lds_doubles(int nmat) {
        return (size_t)nmat * NS * 64 + (size_t)NVAL * W * 16 + (size_t)NTV * DP * 16 + (size_t)16 * LSS + (size_t)2 * W * 16;
    }
    // per-chain global scratch, in vectors of DP doubles: 5 per stack level 2 .. MAXL-1, then the deep sums
    static constexpr int S_DEEP = 5 * (MAXL - 2);
    static constexpr int scratch_slots() { return S_DEEP + (NDEEP * W + DP - 1) / DP; }
};

// Constraint transform of one coordinate, transforms/_constraint.pyx:133-215: the original coordinate xo, the Jacobian J and
// the ratio J2 / J the gradient needs (density.py:749-750).  One exponential serves every kind (lanes of a wave hold
// coordinates of different kinds, so branches would run all of them): with e = exp(-x) the logistic kind is
// t = 1 / (1 + e), J = t (1 - t) rg and J2 / J = -t2 (t2 - 1) / (t2 + 1)^3 / (t (1 - t)) = (e - 1) t  (t2 = exp(x) = 1 / e);
// the one-sided kinds have J2 = J.  Agrees with the statement-by-statement form (bf_to_original, bfhip_eval.h) to rounding.
BF_DEV void bf_to_original_g(double x, int kind, double lo, double rg, double &xo, double &J, double &gj) {
    const double e = bf_exp(kind == 1 ? -x : x);
    const double t = 1. / (1. + e);
    double tmp = x, jt = 1., gq_ = 0.;
    if (kind == 1) { tmp = t; jt = t * (1. - t); gq_ = (e - 1.) * t; }
    if (kind == 2) { tmp = e; jt = e; gq_ = 1.; }
    if (kind == 3) { tmp = 1. - e; jt = -e; gq_ = 1.; }
    xo = lo + tmp * rg;
    J = jt * rg;
    gj = gq_;
}

template <int W, bool NUTS, int FS>
BF_DEV void bf_group_body(const DevModel &m, const SamplerArgs &a, double *lds) {
    constexpr bool DEC = (FS & 2) != 0, TR = (FS & 4) != 0, PLDG = (FS & 8) != 0;
    using G = GroupGeo<W>;
    // (XO: the pipeline form has no S tiles and leaves the x region out: its first region is x - mu)
    constexpr int XO = ((FS & 8) != 0) ? 1 : 0;
    constexpr int DP = G::DP, NS = G::NS, KS = G::KS, NMAT = (DEC ? 3 : 2) - XO, LSS = G::LSS;
    double *XB = lds;                          // [NMAT][NS][64]  B operands: x | x - mu | x_orig - mu_decay
    double *RB = XB + NMAT * NS * 64;          // [NVAL][W][16]   per-wave partial sums
    double *TV = RB + G::NVAL * W * 16;        // [NTV][DP][16]   tree vectors
    double *LS = TV + G::NTV * DP * 16;        // [16][LSS]       subtree stack scalars (one writer: wave 0)
    double *PB = LS + 16 * LSS;                // [2][W][16]      per-wave |x - mu|^2, |x - mu_decay|^2 of the point in flight (proofs)

    const int tid = bf_tid(), lane = tid & 63, j = tid >> 6, c = lane & 15, gq = lane >> 4;
    const int chain = bf_group() * 16 + c;
    const bool real = chain < a.n_chain;
    const bool writer = j == 0 && gq == 0;     // the lane that owns chain c's scalar outputs
    const int d = m.d, dbase = 16 * j + gq;    // element r is dimension dbase + 4 r
    const int nw = a.cfg.n_warmup;
    const double bound_thr = m.use_bound ? m.alpha * m.alpha * (1. - 1e-9) : __builtin_inf();
    const double decay_thr = m.decay_alpha2 * (1. - 1e-9);
#ifndef BF_HOST_EMU
    // pipeline density: its LDS block behind the group's own regions (sixteen-chain layout: the chains are the 16 columns)
    // (its own K-split of the second contraction: 1 -- two or four waves share the jobs here, not sixteen -- and the A operands of
    // both contractions from a row-major copy of C' in LDS, PldLds::CL: bf_group_supports has checked that it fits)
    PldLds PL;
    PldDev plg = m.pld;
    if constexpr (PLDG) {
        plg.KS2 = 1;
        plg.KPJ2 = plg.NS2;
        PL = pld_lds(lds + ((G::lds_doubles(NMAT) + 1) & ~(size_t)1), DP, plg, 16, true, W);
        pld_stage(plg, PL, DP, tid, 64 * W);
    }
#endif

    // ---- constants: A operands of this wave's row tile, per-dimension table rows ----
    double afS[NS], afH[NS], afD[DEC ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        afS[s] = PLDG ? 0. : m.Sf[(j * NS + s) * 64 + lane];
        afH[s] = (PLDG && !m.use_bound) ? 0. : m.Hf[(j * NS + s) * 64 + lane];
        if constexpr (DEC) afD[s] = m.Hdf[(j * NS + s) * 64 + lane];
    }
    double c_lin[4], c_mu[4], c_smu[4], c_hd[4], c_lo[TR ? 4 : 1], c_rg[TR ? 4 : 1], c_dmu[DEC ? 4 : 1], c_hdd[DEC ? 4 : 1];
    int c_kind[TR ? 4 : 1];
    double c_sulo[PLDG ? 4 : 1], c_sudf[PLDG ? 4 : 1], c_pmu[PLDG ? 4 : 1], c_ppr[PLDG ? 4 : 1];   // surrogate input scaling, prior
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int dim = dbase + 4 * r;
        if constexpr (PLDG) {
            const bool su = m.has_su && dim < d;
            c_sulo[r] = su ? m.pd[PD_SU_LO * DP + dim] : 0.;
            c_sudf[r] = su ? m.pd[PD_SU_DIFF * DP + dim] : 1.;
            c_pmu[r] = (m.pld.has_prior && dim < d) ? m.pld.prior_mu[dim] : 0.;
            c_ppr[r] = (m.pld.has_prior && dim < d) ? m.pld.prior_prec[dim] : 0.;
        }
        c_lin[r] = m.pd[PD_LIN * DP + dim];
        c_mu[r] = m.pd[PD_MU * DP + dim];
        c_smu[r] = m.pd[PD_SMU * DP + dim];
        c_hd[r] = m.pd[PD_HD * DP + dim];
        if constexpr (TR) {
            c_kind[r] = (int)m.pd[PD_KIND * DP + dim];
            c_lo[r] = m.pd[PD_LO * DP + dim];
            c_rg[r] = m.pd[PD_RG * DP + dim];
        }
        if constexpr (DEC) { c_dmu[r] = m.pd[PD_DMU * DP + dim]; c_hdd[r] = m.pd[PD_HDD * DP + dim]; }
    }

    // ---- per-chain state: vectors (4 elements per lane) ----
    double q[4], p[4], g[4], var[4], isd[4];           // isd = var^-1/2 (momentum draws)
    double L0p[4], L0q[4], L0g[4];                    // stack level 0: a single waiting leaf
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        q[r] = 0.; p[r] = 0.; g[r] = 0.; var[r] = 1.; isd[r] = 1.;
        L0p[r] = L0q[r] = L0g[r] = 0.;
    }
    // ---- per-chain state: scalars (one copy per lane) ----
    uint64_t rs[4] = {0, 0, 0, 0};
    int mode = M_DONE, i_iter = 0, err = 0;
    int dir = 1, depth = 0, i_leaf = 0, n_prop = 0, diverged = 0, h_accepted = 0;
    double eps = 0., eps_t = 0., start_energy = 0., acc_sum = 0.;
    double T_W = 0., T_acc = 0., T_E = 0., T_logp = 0.;
    double max_de = 0., w_off = 0., tree_W = 1.;
    // An evaluation whose gradient depends on its own sums -- outside the bound (poly.py:496-503; S x_0 of the projected point by
    // linearity, bfhip_oob.h: no pass at x_0) and / or with the decay term active (density.py:744-746) -- is LATE: its second
    // half step, kinetic energy and U-turn sums follow the scalars in a further exchange of the same trip (barrier B3).  A
    // group whose evaluating chains were all late leaves the early exchange of those sums out of its next trip (skip_early:
    // every lane holds every chain's scalars, so the flag is the same in all waves; the numbers do not depend on it).
    bool skip_early = false;
    double L0_W = 0., L0_acc = 0., L0_E = 0., L0_logp = 0.;
    double prop_E = 0., prop_logp = 0.;
    double h_acc = 0., h_de = 0., h_end_E = 0., h_end_logp = 0.;
    double log_step = 0., log_bar = 0., hbar = 0., smu = 0., count = 1., step_now = 0., step_bar = 0.;
    // Welford window counters (metrics.py:186-211): every lane of a chain keeps its own copy in step
    double fg_n = 0., bg_n = 0., n_samples = 0., prev_upd = 0., adapt_window = 0.;
    bool need_E0 = false;     // the running iteration's start energy waits for its kinetic part (this trip's exchange)
    double kin0_part = 0.;
    unsigned long long nlf = 0;
    unsigned int n_trip = 0, n_trip_h = 0, n_trip_late = 0, n_trip_skip = 0;  // measurement: trips of this group, trips that ran the
                                                                              // bound's tiles, had a late exchange, left the early one out

    double *scp = a.sc + (size_t)(real ? chain : 0) * BFHIP_SC_N;
    double *vecp = a.vec + (size_t)(real ? chain : 0) * BFHIP_VEC_N * d;
    double *sbase = a.scratch + ((size_t)(real ? chain : 0) * a.nslot) * DP + dbase;
    double *tvb = TV + dbase * 16 + c;
    double *rbg = a.scratch + ((size_t)(real ? chain : 0) * a.nslot + G::S_DEEP) * DP;  // this chain's deep sums [v][wave]
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
    // tree vector `slot` in LDS (slot may differ from lane to lane)
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
    // subtree stack vector k (0 left p, 1 right p, 2 p_sum, 3 proposal q, 4 proposal gradient) of level lev >= 1
    auto stk_ld = [&](int lev, int k, double (&v)[4]) {
        if (lev == 1) {
            tv_ld(G::T_STK1 + k, v);
        } else {
            const double *sp = sbase + (size_t)((lev - 2) * 5 + k) * DP;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = sp[4 * r];
        }
    };
    auto stk_st = [&](int lev, int k, const double (&v)[4]) {
        if (lev == 1) {
            tv_st(G::T_STK1 + k, v);
        } else {
            double *sp = sbase + (size_t)((lev - 2) * 5 + k) * DP;
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[4 * r] = v[r];
        }
    };
    // sum over this lane's four dimensions, in the butterfly's association (dims 4 apart, then 8 apart)
    auto sum4 = [](const double (&v)[4]) -> double { return (v[0] + v[1]) + (v[2] + v[3]); };
    // sum over g (dims 1 apart, then 2 apart) and post this wave's partial: called by ALL lanes
    auto post = [&](int vi, double part) {
        const double t = bf_xor32_add(bf_xor16_add(part));
        if (gq == 0) RB[(vi * W + j) * 16 + c] = t;
    };
    // N sums posted together: the row swaps of the N values advance side by side, so each step's latency is covered by
    // the other values' instructions (one wave per SIMD: nothing else hides it)
    auto post_n = [&](int vi0, auto &part) {
        constexpr int N = sizeof(part) / sizeof(double);
        // four values take three swaps (bf_pair16_add twice, bf_pair32_add once) and end up one per row g; two values take
        // two; the additions are those of bf_xor32_add(bf_xor16_add(.)) for every value
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
        for (int k = 0; k < NQ; ++k) RB[((vi0 + 4 * k + gq) * W + j) * 16 + c] = sq[k];
        if constexpr (NP > 0) { if (gq < 2) RB[((vi0 + 4 * NQ + gq) * W + j) * 16 + c] = sp[0]; }
        if constexpr (N1 > 0) { if (gq == 0) RB[((vi0 + N - 1) * W + j) * 16 + c] = s1[0]; }
    };
    // N group-wide sums read together (all loads in flight before the first add)
    auto rd_n = [&](int vi0, auto &out) {
        constexpr int N = sizeof(out) / sizeof(double);
        double t[N][W];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) t[i][w2] = RB[((vi0 + i) * W + w2) * 16 + c];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if constexpr (W == 4) out[i] = (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
            else if constexpr (W == 2) out[i] = t[i][0] + t[i][1];
            else out[i] = t[i][0];
        }
    };
    // the group-wide sum: waves 16 dims apart, then 32 apart
    auto sumw = [](const double *rp, int st) -> double {
        if constexpr (W == 4) return (rp[0] + rp[st]) + (rp[2 * st] + rp[3 * st]);
        else if constexpr (W == 2) return rp[0] + rp[st];
        else return rp[0];
    };
    auto rd = [&](int vi) -> double { return sumw(RB + (vi * W) * 16 + c, 16); };
    // sums of merge level lev >= 1, k = 0..5: LDS up to level LSH, this chain's global scratch above
    auto post_lv = [&](int lev, int k, double part) {
        if (lev <= G::LSH) {
            post(G::V_LV + 6 * (lev - 1) + k, part);
        } else {
            const double t = bf_xor32_add(bf_xor16_add(part));
            if (gq == 0 && real) rbg[(6 * (lev - 1 - G::LSH) + k) * W + j] = t;
        }
    };
    auto rd_lv = [&](int lev, int k) -> double {
        if (lev <= G::LSH) return rd(G::V_LV + 6 * (lev - 1) + k);
        return sumw(rbg + (6 * (lev - 1 - G::LSH) + k) * W, 1);
    };
    // metric.random (samplers/hmc_utils/metrics.py:83-86): one xoshiro draw K keys a SplitMix64 counter stream; pair P
    // of the stream gives dimensions 2P (cos) and 2P+1 (sin) by Box-Muller -- the same numbers as the sliced kernel
    // Dimensions d and d ^ 1 of a chain sit in lanes 16 apart (g and g ^ 1, same element r): the even-g lane runs the
    // Box-Muller transform of the pairs of its elements 0 and 1, the odd-g lane of elements 2 and 3, and they trade the
    // halves they do not use (called by all lanes: the exchange is a wave collective; `on` lanes take the result).
    auto draw_momentum = [&](bool on) {
        uint64_t K = 0;
        if (on) K = bf_xoshiro_next(rs);
        const int godd = gq & 1;
        double mine[2], theirs[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 2 * godd + h;                       // the element whose pair this lane computes
            const uint64_t P = (uint64_t)((dbase + 4 * r) >> 1);
            const double u1 = bf_u01_open0(bf_mix64(K + (2 * P + 1) * BF_GOLDEN));
            const double u2 = bf_u01(bf_mix64(K + (2 * P + 2) * BF_GOLDEN));
#if defined(BF_KO_MOM) && !defined(BF_HOST_EMU)
            const double rad = 3.4641016151377544, sn = u1 - 0.5, cs = u2 - 0.5;  // (uniforms of unit variance)
#else
            const double rad = bf_sqrt(-2. * bf_log(u1));
            double sn, cs;
            bf_sincospi(2. * u2, &sn, &cs);
#endif
            mine[h] = godd ? rad * sn : rad * cs;             // odd dimension: sin, even: cos
            theirs[h] = godd ? rad * cs : rad * sn;
        }
        theirs[0] = bf_xor16_get(theirs[0]);
        theirs[1] = bf_xor16_get(theirs[1]);
        if (on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // elements 2 godd, 2 godd + 1 are this lane's own pairs; the other two come from the partner
                const double z = ((r >> 1) == godd) ? mine[r & 1] : theirs[r & 1];
                p[r] = (dbase + 4 * r < d) ? isd[r] * z : 0.;
            }
        }
    };
    // Tree.__init__ (nuts.py:24-43) at the current (q, p, g); the proposal is the starting point
    auto tree_reset = [&]() {
        tree_W = 1.;
        w_off = 0.;
        max_de = 0.;
        depth = 0; acc_sum = 0.; n_prop = 0; diverged = 0; i_leaf = 0;
        eps = (i_iter < nw) ? step_now : step_bar;  // step_size.py:25-29
        dir = 1;
        if (NUTS) dir = (bf_u01(BF_TREE_DRAW(rs)) < 0.5) ? 1 : -1;  // nuts.py:210, log(U) < log(1/2)
        tv_st(G::T_LEFT + 0, q); tv_st(G::T_LEFT + 1, p); tv_st(G::T_LEFT + 2, g);
        tv_st(G::T_RIGHT + 0, q); tv_st(G::T_RIGHT + 1, p); tv_st(G::T_RIGHT + 2, g);
        tv_st(G::T_PROP + 0, q); tv_st(G::T_PROP + 1, g);
        tv_st(G::T_PSUM, p);
        mode = M_LEAF;
    };

    if (real) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rs[k] = a.rng[(size_t)chain * 4 + k];
        log_step = scp[BFHIP_SC_LOG_STEP];
        log_bar = scp[BFHIP_SC_LOG_BAR];
        hbar = scp[BFHIP_SC_HBAR];
        smu = scp[BFHIP_SC_MU];
        count = scp[BFHIP_SC_COUNT];
        fg_n = scp[BFHIP_SC_FG_N];
        bg_n = scp[BFHIP_SC_BG_N];
        n_samples = scp[BFHIP_SC_N_SAMPLES];
        prev_upd = scp[BFHIP_SC_PREV_UPDATE];
        adapt_window = scp[BFHIP_SC_ADAPT_WINDOW];
        step_now = bf_exp(log_step);   // exp(log_step), exp(log_bar): what the statistics report
        step_bar = bf_exp(log_bar);
        i_iter = (int)scp[BFHIP_SC_I_ITER];
        err = (int)scp[BFHIP_SC_ERROR];
        load_vec(BFHIP_VEC_Q, q, 0.);
        load_vec(BFHIP_VEC_VAR, var, 1.);
#pragma unroll
        for (int r = 0; r < 4; ++r) isd[r] = 1. / bf_sqrt(var[r]);  // metrics.py:83-86: kept between the metric's updates
        if (i_iter < a.iter_end && err == 0) mode = M_INIT;
    }
    draw_momentum(mode == M_INIT);

    int trip_no = -1;
    (void)trip_no;
    for (;;) {
        ++trip_no;
        GTRACE(0);
        PTRACE(0);
        if (!bf_any(mode != M_DONE)) break;  // (every wave holds all 16 chains' state: the same decision in all of them)

        // ================= phase A: first half of the leapfrog step, B operands =================
        const bool ev = mode != M_DONE;
        double xs[4], xev[4], jac[4], gj[4], xo[4];
        double ldet[4] = {0., 0., 0., 0.};
        if (ev) {
            eps_t = (mode == M_LEAF) ? eps * (double)dir : 0.;  // compute_state (base_hmc.py:70) is a step of length 0
            const double dt = 0.5 * eps_t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = bf_fma(dt, g[r], p[r]);             // integration.py:80
                q[r] = bf_fma(eps_t, var[r] * p[r], q[r]);  // :82-85
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xs[r] = ev ? q[r] : 0.;
            jac[r] = 1.;
            gj[r] = 0.;
            if constexpr (TR) {
                if (ev) {
                    double J, g2;
                    bf_to_original_g(q[r], c_kind[r], c_lo[r], c_rg[r], xs[r], J, g2);
                    ldet[r] = bf_fabs(J);  // (|dx/dx_t| of this coordinate; its logarithm is taken below, once per lane)
                    jac[r] = J;
                    gj[r] = g2;
                }
            }
            xo[r] = xs[r];
            if constexpr (PLDG) { if (m.has_su) xs[r] = (xs[r] - c_sulo[r]) / c_sudf[r]; }   // the surrogate's input (module.py:76-83)
            xev[r] = xs[r];
            if constexpr (!PLDG) XB[(0 * NS + 4 * j + r) * 64 + lane] = xev[r];
            XB[((1 - XO) * NS + 4 * j + r) * 64 + lane] = xs[r] - c_mu[r];
            if constexpr (DEC) XB[((2 - XO) * NS + 4 * j + r) * 64 + lane] = xo[r] - c_dmu[r];
        }
        // Bound proof.  (x - mu)^T H (x - mu) <= lam_max sum_j hd_j (x_j - mu_j)^2: when that is below alpha^2 for every chain of the
        // group the test of modules/poly.py:467-469 is decided (inside) without the H (x - mu) tiles -- half of the trip's
        // MFMAs.  The partial |x - mu|^2 rides through the barrier the operands need anyway; the outcome is the one the
        // full computation has (margin 1e-9 against its rounding), so results do not depend on whether a group skips.
        if constexpr (!PLDG) {
            double t_r2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double xm = xs[r] - c_mu[r];
                t_r2[r] = ev ? c_hd[r] * (xm * xm) : 0.;   // (the proof's weighted norm: bf_bound_lam_max_weighted)
            }
            double r2p = bf_xor32_add(bf_xor16_add(sum4(t_r2)));
            if (gq == 0) PB[j * 16 + c] = r2p;
            if constexpr (DEC) {
                // the same for the decay term (density.py:740-746): inactive while (x - mu_d)^T H_d (x - mu_d) <= alpha_d^2
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double xm = xo[r] - c_dmu[r];
                    t_r2[r] = ev ? c_hdd[r] * (xm * xm) : 0.;
                }
                double r2d = bf_xor32_add(bf_xor16_add(sum4(t_r2)));
                if (gq == 0) PB[(W + j) * 16 + c] = r2d;
            }
        }
        GTRACE(1);
        PTRACE(1);
        bf_sync();  // B1
        GTRACE(2);
        PTRACE(2);

        // ================= phase B: row tile j of S x, H (x - mu) (, H_decay^T (x - mu_decay)) on MFMA =================
        double sx[4], hv[4], dgr[4];
        bool skipH = false;  // wave-uniform (and the same in every wave): the bound's tiles were left out, b2 is not posted
        bool ranD = false;   // the same for the decay term's tiles
        {
            bf_acc4 aS0 = bf_acc4_zero(), aS1 = bf_acc4_zero(), aH0 = bf_acc4_zero(), aH1 = bf_acc4_zero();
            bf_acc4 aD0 = bf_acc4_zero(), aD1 = bf_acc4_zero();
            constexpr int KH = NS / KS;
            if constexpr (!PLDG) {
#pragma unroll
                for (int s = 0; s < KH; ++s) {
                    aS0 = bf_mfma(afS[s], XB[(0 * NS + s) * 64 + lane], aS0);
                    if constexpr (KS == 2) aS1 = bf_mfma(afS[KH + s], XB[(0 * NS + KH + s) * 64 + lane], aS1);
                }
            }
            double r2 = PLDG ? 0. : PB[c], r2d = (DEC && !PLDG) ? PB[W * 16 + c] : 0.;  // (read behind the S tiles: the matrix pipe is busy with them)
            if constexpr (!PLDG) {
#pragma unroll
                for (int w2 = 1; w2 < W; ++w2) {
                    r2 += PB[w2 * 16 + c];
                    if constexpr (DEC) r2d += PB[(W + w2) * 16 + c];
                }
            }
            const bool inside = !PLDG && !a.no_bound_proof && m.lam_max * r2 < bound_thr;  // (NaN: not proven; the pipeline form has no proof)
            skipH = PLDG ? !m.use_bound : !bf_any(!inside);
            n_trip += 1;
            n_trip_h += skipH ? 0 : 1;
            if (!skipH) {
#pragma unroll
                for (int s = 0; s < KH; ++s) {
                    aH0 = bf_mfma(afH[s], XB[((1 - XO) * NS + s) * 64 + lane], aH0);
                    if constexpr (KS == 2) aH1 = bf_mfma(afH[KH + s], XB[((1 - XO) * NS + KH + s) * 64 + lane], aH1);
                }
            }
            if constexpr (DEC) {
                const bool calm = !PLDG && !a.no_bound_proof && m.lam_max_d * r2d < decay_thr;
                ranD = bf_any(!calm);
                if (ranD) {
#pragma unroll
                    for (int s = 0; s < KH; ++s) {
                        aD0 = bf_mfma(afD[s], XB[((2 - XO) * NS + s) * 64 + lane], aD0);
                        if constexpr (KS == 2) aD1 = bf_mfma(afD[KH + s], XB[((2 - XO) * NS + KH + s) * 64 + lane], aD1);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sx[r] = (KS == 2) ? aS0[r] + aS1[r] : aS0[r];
                hv[r] = (KS == 2) ? aH0[r] + aH1[r] : aH0[r];
                dgr[r] = DEC ? ((KS == 2) ? aD0[r] + aD1[r] : aD0[r]) : 0.;
            }
        }

        GTRACE(3);
        PTRACE(3);
        // ================= phase C: the evaluation's sums, and the U-turn sums of the leaf it completes =================
        double ge[4], pn[4];
        const double dt_c = 0.5 * eps_t;
        const bool early = PLDG ? false : !skip_early;   // the kinetic energy and the U-turn sums ride in this exchange (same in all waves)
        if constexpr (PLDG) {
            // the sums that do not need the contractions: the bound's and the decay term's radii, the log-Jacobian, the prior
            double t_b2[4], t_bd2[4], t_pr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t_b2[r] = (xs[r] - c_mu[r]) * hv[r];
                t_bd2[r] = DEC ? (xo[r] - c_dmu[r]) * dgr[r] : 0.;
                const double dx = ev ? xo[r] - c_pmu[r] : 0.;
                t_pr[r] = c_ppr[r] * dx * dx;
            }
            if (!skipH) post(G::V_B2, sum4(t_b2));
            if constexpr (DEC) post(G::V_BD2, sum4(t_bd2));
            if (m.pld.has_prior) post(G::V_VAL, sum4(t_pr));
            if constexpr (TR) {   // sum_r log|J_r| as the logarithm of the lane's product (see the polynomial form below)
                const double pj = (ldet[0] * ldet[1]) * (ldet[2] * ldet[3]);
                double lsum;
                if (bf_any(ev && !(pj > 1e-280 && pj < 1e280))) {
                    double l4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) l4[r] = ev ? bf_log(ldet[r]) : 0.;
                    lsum = sum4(l4);
                } else {
                    lsum = ev ? bf_log(pj) : 0.;
                }
                post(G::V_LOGDET, lsum);
            }
            if (bf_any(need_E0)) post(G::V_KIN0, kin0_part);
#pragma unroll
            for (int r = 0; r < 4; ++r) { ge[r] = 0.; pn[r] = p[r]; }
        } else {
            double gn[4], t_val[4], t_b2[4], t_bd2[4], t_kin[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gn[r] = sx[r] + c_lin[r];
                // value of the linear + quadratic surrogate per dimension: lin_r x_r + x_r (S x)_r / 2
                t_val[r] = bf_fma(0.5 * xev[r], sx[r], c_lin[r] * xev[r]);
                const double xm = xs[r] - c_mu[r];
                t_b2[r] = xm * hv[r];
                t_bd2[r] = DEC ? (xo[r] - c_dmu[r]) * dgr[r] : 0.;
                // inside the bound (and the decay ellipsoid) the gradient is complete: chain rule, transform term
                // (module.py:226, density.py:558,747-750), second half of the step (integration.py:90) and the kinetic
                // energy (metrics.py:88-91) ride along
                double t = gn[r];
                if constexpr (TR) t = t * jac[r];
                if constexpr (TR) t += gj[r];
                ge[r] = t;
                pn[r] = bf_fma(dt_c, ge[r], p[r]);
                t_kin[r] = pn[r] * (var[r] * pn[r]);
            }
            static_assert(G::V_KIN == 0 && G::V_VAL == 1 && G::V_B2 == 2, "posted as one batch");
            if (skipH) {
                if (early) {
                    double e2[2] = {sum4(t_kin), sum4(t_val)};
                    post_n(G::V_KIN, e2);
                } else {
                    post(G::V_VAL, sum4(t_val));
                }
            } else if (early) {
                double e3[3] = {sum4(t_kin), sum4(t_val), sum4(t_b2)};
                post_n(G::V_KIN, e3);
            } else {
                double e2[2] = {sum4(t_val), sum4(t_b2)};
                post_n(G::V_VAL, e2);
            }
            if constexpr (DEC) post(G::V_BD2, sum4(t_bd2));
            if constexpr (TR) {
                // sum_r log|J_r| (density.py:748) as the logarithm of the lane's product: one log instead of four.  The
                // product of four Jacobians stays far from underflow unless a coordinate sits ~700 logistic units deep
                // in a bound's tail; then (any lane of the wave) the four logarithms are taken one by one.
                double pj = (ldet[0] * ldet[1]) * (ldet[2] * ldet[3]);
                double lsum;
                if (bf_any(ev && !(pj > 1e-280 && pj < 1e280))) {
                    double l4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) l4[r] = ev ? bf_log(ldet[r]) : 0.;
                    lsum = sum4(l4);
                } else {
                    lsum = ev ? bf_log(pj) : 0.;
                }
                post(G::V_LOGDET, lsum);
            }
            if (bf_any(need_E0)) post(G::V_KIN0, kin0_part);
        }
        // U-turn sums for the subtrees that the leaf in flight completes (nuts.py:146-161 per merge, :88-101 per doubling).
        // tTL / tTPs: the merged subtree's first momentum and p_sum; tPS: the tree's p_sum after the doubling.
        double tTL[4], tTPs[4], tPS[4];
        int nm = 0;  // number of merge levels: trailing one bits of i_leaf, at most depth
        bool any_m0 = false, any_lv1 = false, any_ext = false;  // which groups of sums this trip carries (wave-uniform)
        auto uturn_sums = [&](bool act) {
            // act: this lane's chain takes part (all lanes run the collectives)
            nm = 0;
            if (act) {  // trailing one bits of i_leaf, at most depth
                const int t1 = __builtin_ctz(~(unsigned)i_leaf);
                nm = t1 < depth ? t1 : depth;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { tTL[r] = pn[r]; tTPs[r] = pn[r]; tPS[r] = 0.; }
            any_m0 = bf_any(act && nm >= 1);
            any_lv1 = bf_any(act && nm >= 2);
            any_ext = bf_any(act && nm == depth);
            if (any_m0) {  // level 0: the waiting leaf L0 and the new one (nuts.py:150-151; no sub-span checks)
                double t0[4], t1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double ps0 = L0p[r] + pn[r];
                    t0[r] = ps0 * (var[r] * L0p[r]);
                    t1[r] = ps0 * (var[r] * pn[r]);
                    if (act && nm >= 1) { tTPs[r] = ps0; tTL[r] = L0p[r]; }
                }
                double m2[2] = {sum4(t0), sum4(t1)};
                post_n(G::V_M0, m2);
            }
            for (int lev = 1; bf_any(act && lev < nm); ++lev) {
                const bool on = act && lev < nm;
                double A[4] = {0., 0., 0., 0.}, B[4] = {0., 0., 0., 0.}, S1[4] = {0., 0., 0., 0.};
                if (on) { stk_ld(lev, 0, A); stk_ld(lev, 1, B); stk_ld(lev, 2, S1); }
                double t[6][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double psum = S1[r] + tTPs[r];
                    const double vA = var[r] * A[r], vB = var[r] * B[r], vC = var[r] * tTL[r], vD = var[r] * pn[r];
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
                    post_n(G::V_LV + 6 * (lev - 1), s6);
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) post_lv(lev, k, sum4(t[k]));
                }
            }
            if (any_ext) {  // the doubling completes: Tree.extend's checks, nuts.py:86-101
                const bool on = act && nm == depth;
                double PS[4] = {0., 0., 0., 0.}, Lp[4] = {0., 0., 0., 0.}, Rp[4] = {0., 0., 0., 0.};
                if (on) { tv_ld(G::T_PSUM, PS); tv_ld(G::T_LEFT + 1, Lp); tv_ld(G::T_RIGHT + 1, Rp); }
                double t[6][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double ps = PS[r] + tTPs[r];  // :86 (in place)
                    const double vN = var[r] * pn[r], vT = var[r] * tTL[r], vL = var[r] * Lp[r], vR = var[r] * Rp[r];
                    // NOTE (reference behaviour, kept on purpose): leftmost_p_sum (dir > 0) / rightmost_p_sum (dir < 0)
                    // alias self.p_sum, which line 86 has just updated in place.
                    if (dir > 0) {
                        const double ps1 = ps + tTL[r], ps2 = Rp[r] + tTPs[r];
                        t[0][r] = ps * vL; t[1][r] = ps * vN; t[2][r] = ps1 * vL; t[3][r] = ps1 * vT;
                        t[4][r] = ps2 * vR; t[5][r] = ps2 * vN;
                    } else {
                        const double ps1 = tTPs[r] + Lp[r], ps2 = tTL[r] + ps;
                        t[0][r] = ps * vN; t[1][r] = ps * vR; t[2][r] = ps1 * vN; t[3][r] = ps1 * vL;
                        t[4][r] = ps2 * vT; t[5][r] = ps2 * vR;
                    }
                    if (on) tPS[r] = ps;
                }
                double s6[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) s6[k] = sum4(t[k]);
                post_n(G::V_EXT, s6);
            }
        };
        // every chain whose leaf this evaluation completes
        const bool spec1 = NUTS && ev && mode == M_LEAF;
        GTRACE(4);
        if (NUTS && early) uturn_sums(spec1);
        GTRACE(5);
        PTRACE(4);
        bf_sync();  // B2
        GTRACE(6);
        PTRACE(5);

        // ================= the evaluation's scalars =================
        // this trip's sums, fetched together (one LDS round trip instead of one per value)
        double s_kin = 0., s_val = 0., s_b2 = 0.;
        bool late_sync_done = false;
        double sv_k0[1] = {0.}, sv_m0[2] = {1., 1.}, sv_l1[6] = {1., 1., 1., 1., 1., 1.}, sv_x[6] = {1., 1., 1., 1., 1., 1.};
        if constexpr (PLDG) {
            // (the pipeline form reads its sums below)
        } else if (skipH) {   // inside the bound, proven
            if (early) {
                double sv_2[2];
                rd_n(G::V_KIN, sv_2);
                s_kin = sv_2[0]; s_val = sv_2[1];
            } else {
                s_val = rd(G::V_VAL);
            }
        } else if (early) {
            double sv_3[3];
            rd_n(G::V_KIN, sv_3);
            s_kin = sv_3[0]; s_val = sv_3[1]; s_b2 = sv_3[2];
        } else {
            double sv_2[2];
            rd_n(G::V_VAL, sv_2);
            s_val = sv_2[0]; s_b2 = sv_2[1];
        }
        const bool any_e0 = bf_any(need_E0);
        if (any_e0) rd_n(G::V_KIN0, sv_k0);
        // the U-turn sums of this exchange (the flags are uturn_sums' own: wave-uniform)
        auto read_uturn = [&]() {
            if (any_m0) rd_n(G::V_M0, sv_m0);
            if (any_lv1) rd_n(G::V_LV, sv_l1);
            if (any_ext) rd_n(G::V_EXT, sv_x);
        };
        if (NUTS && early) read_uturn();
        bool fin = false, late = false, dec_on = false;
        double logp_new = 0., kin = 0.;
        double f = (m.c0 + s_val) + 0.;
#ifndef BF_HOST_EMU
        if constexpr (PLDG) {
            // ================= the pipeline density (bfhip_pld.h; core/density.py:527-560, modules/poly.py:430-503) =================
            const PldDev &pl = plg;
            const double r_b2 = skipH ? 0. : rd(G::V_B2);
            const double r_pr = pl.has_prior ? rd(G::V_VAL) : 0.;
            double r_bd2 = 0.;
            if constexpr (DEC) r_bd2 = rd(G::V_BD2);
            // the bound is decided FIRST, so a point outside the ellipsoid is evaluated once, at its projection (poly.py:480-503)
            double beta_o = 0.;
            if (!skipH && !(r_b2 < m.alpha * m.alpha * (1. - 1e-12))) {   // :467-469
                const double b = bf_sqrt(r_b2);
                if (b > m.alpha) beta_o = b;
            }
            {
                double *xe = PL.XE + c * (DP + 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dim = dbase + 4 * r;
                    const double x_ev = beta_o > 0. ? (m.alpha * xs[r] + (beta_o - m.alpha) * c_mu[r]) / beta_o : xs[r];   // :482
                    xe[dim] = (ev && dim < d) ? x_ev : 0.;
                }
                if (writer) {
                    xe[DP] = 1.;
                    xe[DP + 1] = 0.;
                    PL.CH[c] = ev ? beta_o : 0.;
                }
            }
            PTRACE(6);
            bf_sync();  // P0: the evaluation points of the 16 chains
            PTRACE(7);
            {   // the monomials of every chain, the B operand of GEMM1: lane (c, gq) of wave j takes monomials 4 j + gq, + 4 W, ... of
                // chain c, four at a time (the table word and the three factors of a monomial are dependent LDS reads)
                const double *xe2 = PL.XE + c * (DP + 2);
                for (int p0 = 4 * j + gq; p0 < pl.PP; p0 += 16 * W) {
                    unsigned mo[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) mo[u] = PL.MONO[p0 + 4 * W * u < pl.PP ? p0 + 4 * W * u : p0];
                    double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (xe2[mo[u] & 255u] * xe2[(mo[u] >> 8) & 255u]) * xe2[(mo[u] >> 16) & 255u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int pm = p0 + 4 * W * u;
                        if (pm < pl.PP) PL.PHI[(pm >> 2) * PLD_XS + c + 16 * (pm & 3)] = v[u];
                    }
                }
            }
            PTRACE(8);
            bf_sync();  // P1
            PTRACE(9);
            pld_gemm1_cl(pl, PL, m.alpha, j, W, lane);
            PTRACE(10);
            bf_sync();  // P2: residuals
            PTRACE(11);
            pld_gemm2_cl(pl, PL, j, W, lane);
            PTRACE(12);
            bf_sync();  // P3: W = C'^T r
            PTRACE(13);
            double s_rr = 0., s_fr = 0.;
#pragma unroll
            for (int w2 = 0; w2 < W; ++w2) {
                s_rr += PL.RED[(w2 * 2 + 0) * 16 + c];
                s_fr += PL.RED[(w2 * 2 + 1) * 16 + c];
            }
            // (J_0^T r) of this lane's four dimensions: pld_grad's sum, entry by entry, the four dimensions side by side -- an entry is
            // a chain of three dependent LDS reads, and four gathers one after the other were a third of the trip
            double gj0[4] = {0., 0., 0., 0.};
            {
                const double *xe = PL.XE + c * (DP + 2);
                unsigned long long en[4];   // (the table words of the NEXT entry are on their way while one is gathered)
#pragma unroll
                for (int r = 0; r < 4; ++r) en[r] = pl.n_ent > 0 ? PL.GT[dbase + 4 * r] : 0ull;
                for (int i = 0; i < pl.n_ent; ++i) {
                    unsigned long long nx[4];
                    const int i1 = i + 1 < pl.n_ent ? i + 1 : i;
#pragma unroll
                    for (int r = 0; r < 4; ++r) nx[r] = PL.GT[(size_t)i1 * DP + dbase + 4 * r];
                    double wv[4], cf[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned eh = (unsigned)(en[r] >> 32);
                        const int pm = (int)(unsigned)en[r], off = (pm >> 2) * PLD_XS + c + 16 * (pm & 3);
                        wv[r] = PL.PHI[off];   // (K-split 1: one slot)
                        cf[r] = xe[eh & 255u] * xe[(eh >> 8) & 255u];
                        wv[r] = (double)((eh >> 16) & 255u) * wv[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { gj0[r] += wv[r] * cf[r]; en[r] = nx[r]; }
                }
            }
            if (beta_o > 0.) {   // (compressed outputs: the tails of Q^T f_mu' and Q^T y' as scalars, bfhip_pipeline_upload)
                const double b = (beta_o - m.alpha) / m.alpha;
                s_rr += b * (b * pl.k_ff + 2. * pl.k_fy);
                s_fr += b * pl.k_ff + pl.k_fy;
            }
            if (bf_any(ev && beta_o > 0.)) {   // modules/poly.py:494-496, contracted with r: one more sum over the dimensions
                double t_dj[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) t_dj[r] = gj0[r] * (xs[r] - c_mu[r]);
                post(G::V_VAL, sum4(t_dj));
                bf_sync();
                const double r_dotj = rd(G::V_VAL);
                if (beta_o > 0.) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) gj0[r] += (s_fr / m.alpha - r_dotj / beta_o) * (hv[r] / beta_o);
                }
            }
            f = pl.logp0 - 0.5 * s_rr;
            if (pl.has_prior) f += pl.prior_c0 - 0.5 * r_pr;   // the last module: like + log prior of the original-space inputs
            if constexpr (DEC) {   // density.py:740-746
                const double ex = r_bd2 - m.decay_alpha2;
                f -= m.decay_gamma * (ex > 0. ? ex : (ex != ex ? ex : 0.));
                dec_on = r_bd2 > m.decay_alpha2;
            }
            if constexpr (TR) f += rd(G::V_LOGDET);   // :748
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double t = -gj0[r];                         // density.py:552-560: dot(J_like, J_surrogate)
                if (m.has_su) t = t / c_sudf[r];            // module.py:226
                t = t * jac[r];                             // density.py:558
                if (pl.has_prior) t += -(c_ppr[r] * (xo[r] - c_pmu[r])) * jac[r];
                if constexpr (DEC) { if (dec_on) t -= 2. * m.decay_gamma * dgr[r]; }
                if constexpr (TR) t += gj[r];               // :749-750
                ge[r] = t;
                pn[r] = bf_fma(dt_c, ge[r], p[r]);          // integration.py:90
            }
            late = ev;
            PTRACE(14);
        }
#endif
        // (the common trip -- every chain proven inside the bound, no decay term -- skips the rare branches as one: per-lane
        // branches cost a round trip through the scalar unit each, with one wave per SIMD nothing hides it)
        const bool rare = !PLDG && (DEC || !skipH);   // wave-uniform
        if (rare) {
            // beta = sqrt(b2) is only needed outside the ellipsoid; the test beta > alpha (poly.py:467-469) is decided on the
            // squares whenever b2 is not within rounding distance of alpha^2
            double bt = 0.;
            bool any_oob = false, any_dec = false;
            if (!skipH) {
                const double a2 = m.alpha * m.alpha;
                if (ev && !(s_b2 < a2 * (1. - 1e-12))) bt = bf_sqrt(s_b2);
                any_oob = bf_any(ev && bt > m.alpha);
            }
            if constexpr (DEC) {  // density.py:740-746
                if (ev) {
                    const double r_bd2 = rd(G::V_BD2);
                    const double ex = r_bd2 - m.decay_alpha2;
                    f -= m.decay_gamma * (ex > 0. ? ex : (ex != ex ? ex : 0.));
                    dec_on = r_bd2 > m.decay_alpha2;
                }
                any_dec = bf_any(dec_on);
            }
            if (__builtin_expect(any_oob || any_dec, 0)) {   // (cold: the register allocator spills here, not in the common trip)
                // Some chain's gradient depends on the evaluation's own sums.  Rare in this layout, so nothing of phase B is kept
                // for it: the trip's tiles run again on the operands that are still in XB (the same numbers), and outside the
                // bound the two sums of the extrapolation (bfhip_oob.h: a1 = (x - mu) . (S mu + lin), a2 = (x - mu) . S (x - mu)) take
                // an exchange of their own, through the slots of the value and the bound.
                double sx2[4], hv2[4], dgr2[4];
                {
                    bf_acc4 aS0 = bf_acc4_zero(), aS1 = bf_acc4_zero(), aH0 = bf_acc4_zero(), aH1 = bf_acc4_zero();
                    bf_acc4 aD0 = bf_acc4_zero(), aD1 = bf_acc4_zero();
                    constexpr int KH = NS / KS;
#pragma unroll
                    for (int s2 = 0; s2 < KH; ++s2) {
                        aS0 = bf_mfma(afS[s2], XB[(0 * NS + s2) * 64 + lane], aS0);
                        if constexpr (KS == 2) aS1 = bf_mfma(afS[KH + s2], XB[(0 * NS + KH + s2) * 64 + lane], aS1);
                    }
                    if (!skipH) {
#pragma unroll
                        for (int s2 = 0; s2 < KH; ++s2) {
                            aH0 = bf_mfma(afH[s2], XB[(1 * NS + s2) * 64 + lane], aH0);
                            if constexpr (KS == 2) aH1 = bf_mfma(afH[KH + s2], XB[(1 * NS + KH + s2) * 64 + lane], aH1);
                        }
                    }
                    if constexpr (DEC) {
                        if (ranD) {
#pragma unroll
                            for (int s2 = 0; s2 < KH; ++s2) {
                                aD0 = bf_mfma(afD[s2], XB[(2 * NS + s2) * 64 + lane], aD0);
                                if constexpr (KS == 2) aD1 = bf_mfma(afD[KH + s2], XB[(2 * NS + KH + s2) * 64 + lane], aD1);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sx2[r] = (KS == 2) ? aS0[r] + aS1[r] : aS0[r];
                        hv2[r] = (KS == 2) ? aH0[r] + aH1[r] : aH0[r];
                        dgr2[r] = DEC ? ((KS == 2) ? aD0[r] + aD1[r] : aD0[r]) : 0.;
                    }
                }
                double s_a1 = 0., s_a2 = 0.;
                if (any_oob) {
                    bf_sync();  // the early exchange's sums have been read by every wave
                    double t_a1[4], t_a2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double xm = XB[(1 * NS + 4 * j + r) * 64 + lane];   // x - mu, this lane's own operand
                        t_a1[r] = xm * (c_smu[r] + c_lin[r]);
                        t_a2[r] = xm * (sx2[r] - c_smu[r]);
                    }
                    double e2[2] = {sum4(t_a1), sum4(t_a2)};
                    post_n(G::V_VAL, e2);
                    bf_sync();
                    double sv_2[2];
                    rd_n(G::V_VAL, sv_2);
                    s_a1 = sv_2[0]; s_a2 = sv_2[1];
                }
                if (ev) {
                    double gl[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) gl[r] = sx2[r] + c_lin[r];
                    if (bt > m.alpha) {   // outside the alpha-ellipsoid (poly.py:480-503)
                        const BfOob o = bf_oob_scalars(m.alpha, m.inv_alpha, m.f_mu, m.f_poly_mu, bt, s_a1, s_a2);
                        f = o.f;
                        if constexpr (DEC) {   // (the decay term was taken off the value at x above: off this one again)
                            const double r_bd2 = rd(G::V_BD2);
                            const double ex = r_bd2 - m.decay_alpha2;
                            f -= m.decay_gamma * (ex > 0. ? ex : (ex != ex ? ex : 0.));
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) gl[r] = bf_oob_grad(o, c_smu[r] + c_lin[r], sx2[r] - c_smu[r], hv2[r]);
                        late = true;
                    }
                    late = late || dec_on;
                    if (late) {   // the complete gradient: poly.py:496-503, chain rule, decay, transform term; second half step
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double t = gl[r];
                            if constexpr (TR) t = t * jac[r];
                            if constexpr (DEC) { if (dec_on) t -= 2. * m.decay_gamma * dgr2[r]; }
                            if constexpr (TR) t += gj[r];
                            ge[r] = t;
                            pn[r] = bf_fma(dt_c, ge[r], p[r]);
                        }
                    }
                }
                late_sync_done = any_oob;
            }
        }
        if constexpr (TR && !PLDG) f += rd(G::V_LOGDET);
        if (ev) {
            fin = true;
            logp_new = f;
            kin = s_kin;
        }
        // the late exchange: the kinetic energy and the U-turn sums with the complete momenta (all chains of the group again --
        // the ones that were not late post the numbers they posted before)
        const bool late_x = bf_any(ev && (late || !early));
        n_trip_late += late_x ? 1 : 0;
        n_trip_skip += early ? 0 : 1;
        if (__builtin_expect(late_x, PLDG ? 1 : 0)) {   // (the pipeline form's evaluations are always late: not a cold path there)
            if (early && !late_sync_done) bf_sync();  // B2': the early exchange's sums have been read by every wave
            double t_kin[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) t_kin[r] = pn[r] * (var[r] * pn[r]);
            post(G::V_KIN, sum4(t_kin));
            if (NUTS) uturn_sums(spec1);
            bf_sync();  // B3
            kin = rd(G::V_KIN);
            if (NUTS) read_uturn();
        }
        skip_early = late_x && !bf_any(ev && !late);
        double E_new = 0.;
        if (fin) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[r] = pn[r]; g[r] = ge[r]; }  // integration.py:90
            E_new = 0.5 * kin - logp_new;                                  // :92-93
        }
        // the start energy of an iteration that began at the end of the previous trip
        if (any_e0 && need_E0 && ev) {
            const double E0 = 0.5 * sv_k0[0] - prop_logp;  // integration.py:28-34
            if (!(bf_fabs(E0) <= BF_DBL_MAX)) err = 1;           // base_hmc.py:72-76
            start_energy = E0;
            prop_E = E0;
            need_E0 = false;
        }

        GTRACE(7);
        PTRACE(15);
        GTRACE2(1);
        // ================= per-chain state machine =================
        enum { S_NONE, S_MERGE, S_ABORT, S_DBL_END, S_END };
        int st = S_NONE, lev = 0, src = -1;  // src: whose proposal the finished subtree holds (-1 this leaf, 0 L0, l stack level l)
        double dE = 0., aw = 0., sc_ = 1.;
        bool resc = false;
        const bool ok = fin && err == 0;
        const bool is_init = ok && mode == M_INIT, leaf = ok && mode == M_LEAF;
        if (bf_any(is_init)) {  // the first iteration of a launch only
            if (is_init) {
                // BaseHMC.astep start: base_hmc.py:70-76
                if (!(bf_fabs(E_new) <= BF_DBL_MAX)) {
                    err = 1;
                } else {
                    start_energy = E_new;
                    prop_E = E_new;
                    prop_logp = logp_new;
                    tree_reset();
                }
            }
        }
        if (leaf) {
            nlf += 1;
            if constexpr (NUTS) {
                // ---- Tree._single_step: nuts.py:105-132 (selects instead of branches: see above) ----
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
                // multinomial weight exp(-dE) relative to a running offset w_off (exact streaming log-sum-exp)
                aw = dv ? 0. : -dE - w_off;
                resc = aw > 600.;
            } else {
                // ---- HMC._hamiltonian_step: samplers/hmc.py:16-49 ----
                i_leaf += 1;
                if (i_leaf >= a.cfg.n_int_step) {
                    const bool finite = bf_fabs(E_new) <= BF_DBL_MAX;
                    const double h_dE = finite ? (start_energy - E_new) : -__builtin_inf();
                    diverged = (!finite || bf_fabs(h_dE) > a.cfg.max_change) ? 1 : 0;
                    double h_accept_stat = bf_exp(h_dE);
                    if (h_accept_stat > 1.) h_accept_stat = 1.;
                    h_accepted = 0;
                    if (!diverged) h_accepted = !(bf_u01(bf_xoshiro_next(rs)) >= h_accept_stat);
                    if (h_accepted) {
                        tv_st(G::T_PROP + 0, q);
                        tv_st(G::T_PROP + 1, g);
                        prop_logp = logp_new;
                    }
                    h_de = h_dE;
                    h_acc = h_accept_stat;
                    h_end_E = E_new;
                    h_end_logp = logp_new;
                    st = S_END;
                }
            }
        }
        GTRACE2(2);
        if constexpr (NUTS) {
            if (bf_any(resc)) {  // (rare) the weights follow a new offset; the stacked subtrees' are wave 0's
                if (resc) {
                    sc_ = bf_exp(-aw);
                    tree_W = tree_W * sc_;
                    L0_W *= sc_;
                    w_off = w_off + aw;
                    aw = 0.;
                }
                bf_sync();
                if (writer && resc)
                    for (int l2 = 1; l2 < depth; ++l2) lsc[l2 * 4 + LS_LS] *= sc_;
                bf_sync();
            }
            if (st == S_MERGE) {
                T_W = BF_LEAF_EXP(aw);
                const double pacc = (w_off == 0.) ? T_W : BF_LEAF_EXP(-dE);
                T_acc = pacc > 1. ? 1. : pacc;
                GTRACE2(3);
                if (nm >= 1) {
                    // ---- level-0 merge with the waiting leaf L0 (nuts.py:146-178) ----
                    const double d0 = sv_m0[0], d1 = sv_m0[1];
                    T_acc = L0_acc + T_acc;  // :173
                    const double Wsum = L0_W + T_W;
                    if (Wsum != Wsum) err = 2;
                    const double u = bf_u01(BF_TREE_DRAW(rs));  // :163-167, drawn even when turning
                    lev = 1;
                    if ((d0 <= 0.) || (d1 <= 0.)) {
                        st = S_ABORT;
                    } else {
                        // logbern(ls2 - logaddexp(ls1, ls2))  <=>  U * (W1 + W2) < W2
                        if (!((u * Wsum < T_W) || (u == 0.))) { src = 0; T_E = L0_E; T_logp = L0_logp; }
                        T_W = Wsum;
                    }
                }
                // ---- merge upwards while the finished subtree is a right child ----
                while (st == S_MERGE && lev < nm) {
                    bool turning = false;
                    if (lev == 1) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) turning = turning || (sv_l1[k] <= 0.);
                    } else {
#pragma unroll
                        for (int k = 0; k < 6; ++k) turning = turning || (rd_lv(lev, k) <= 0.);
                    }
                    const double *lsp = lsc + lev * 4;
                    T_acc = lsp[LS_ACC] + T_acc;  // :173
                    const double Wsum = lsp[LS_LS] + T_W;
                    if (Wsum != Wsum) err = 2;
                    const double u = bf_u01(BF_TREE_DRAW(rs));  // consumed even when this merge's check says turning
                    const bool keep_t2 = (u * Wsum < T_W) || (u == 0.);
                    if (turning) {
                        st = S_ABORT;  // ancestors above this level still add their accept sums
                    } else {
                        if (!keep_t2) { src = lev; T_E = lsp[LS_E]; T_logp = lsp[LS_LOGP]; }
                        T_W = Wsum;
                    }
                    lev += 1;
                }
                if (st == S_MERGE) {
                    if (lev < depth) {
                        // the subtree waits for its right sibling
                        if (lev == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { L0p[r] = p[r]; L0q[r] = q[r]; L0g[r] = g[r]; }
                            L0_W = T_W; L0_acc = T_acc; L0_E = E_new; L0_logp = logp_new;
                        } else {
                            double tq[4], tg[4];
                            if (src < 0) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) { tq[r] = q[r]; tg[r] = g[r]; }
                            } else if (src == 0) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) { tq[r] = L0q[r]; tg[r] = L0g[r]; }
                            } else {
                                stk_ld(src, 3, tq);
                                stk_ld(src, 4, tg);
                            }
                            stk_st(lev, 0, tTL); stk_st(lev, 1, p); stk_st(lev, 2, tTPs); stk_st(lev, 3, tq); stk_st(lev, 4, tg);
                            if (writer) {
                                double *lsp = lsc + lev * 4;
                                lsp[LS_LS] = T_W; lsp[LS_ACC] = T_acc; lsp[LS_E] = T_E; lsp[LS_LOGP] = T_logp;
                            }
                        }
                        i_leaf += 1;
                        st = S_NONE;
                    } else {
                        st = S_DBL_END;
                    }
                }
            }
        }
        GTRACE2(4);
        // nothing below happens in a trip in which no chain of the group ends a subtree's doubling, a tree or an iteration
        const bool any_end = bf_any(st != S_NONE);
        if (any_end) {
        if constexpr (NUTS) {
            if (st == S_ABORT) {
                // unwind: every pending ancestor adds its left half's accept_sum (nuts.py:173)
                for (int al = (diverged ? 0 : lev); al < depth; ++al)
                    if ((i_leaf >> al) & 1) T_acc = (al == 0 ? L0_acc : lsc[al * 4 + LS_ACC]) + T_acc;
                depth += 1;  // nuts.py:71-73
                acc_sum += T_acc;
                st = S_END;
            } else if (st == S_DBL_END) {
                // ---- Tree.extend after a complete subtree: nuts.py:71-103 ----
                depth += 1;
                acc_sum += T_acc;
                {   // :81-83  logbern(ls_new - ls_old)  <=>  U * W_old < W_new
                    if (T_W != T_W || tree_W != tree_W) err = 2;
                    const double u = bf_u01(BF_TREE_DRAW(rs));
                    if ((u * tree_W < T_W) || (u == 0.)) {
                        if (src < 0) {
                            tv_st(G::T_PROP + 0, q);
                            tv_st(G::T_PROP + 1, g);
                        } else if (src == 0) {
                            tv_st(G::T_PROP + 0, L0q);
                            tv_st(G::T_PROP + 1, L0g);
                        } else {
                            double tq[4], tg[4];
                            stk_ld(src, 3, tq);
                            stk_ld(src, 4, tg);
                            tv_st(G::T_PROP + 0, tq);
                            tv_st(G::T_PROP + 1, tg);
                        }
                        prop_E = T_E;
                        prop_logp = T_logp;
                    }
                    tree_W = tree_W + T_W;  // :85
                }
                bool turning = false;
#pragma unroll
                for (int k = 0; k < 6; ++k) turning = turning || (sv_x[k] <= 0.);
                tv_st(G::T_PSUM, tPS);
                {
                    const int eo = (dir > 0) ? G::T_RIGHT : G::T_LEFT;
                    tv_st(eo + 0, q); tv_st(eo + 1, p); tv_st(eo + 2, g);
                }
                if (turning || depth >= a.cfg.max_treedepth) {
                    st = S_END;
                } else {
                    const int nd = (bf_u01(BF_TREE_DRAW(rs)) < 0.5) ? 1 : -1;  // nuts.py:210
                    if (nd != dir) {
                        const int eo = (nd > 0) ? G::T_RIGHT : G::T_LEFT;
                        tv_ld(eo + 0, q); tv_ld(eo + 1, p); tv_ld(eo + 2, g);
                    }
                    dir = nd;
                    i_leaf = 0;
                    st = S_NONE;
                }
            }
        }
        GTRACE2(5);
        // ================= iteration end: base_hmc.py:80-85 =================
        bool new_iter = false;
        if (st == S_END && err == 0) {
            const bool warm = i_iter < nw;
            const double accept_stat = NUTS ? acc_sum / (double)n_prop : h_acc;  // nuts.py:186
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
            // the proposal is the new sample and the start of the next iteration (value and gradient came with it)
            tv_ld(G::T_PROP + 0, q);
            tv_ld(G::T_PROP + 1, g);
            const int orow = i_iter - a.iter_out0;
#if defined(BF_KO_STATS) && !defined(BF_HOST_EMU)
            if (false) {
#else
            if (orow >= 0 && orow < a.n_out) {
#endif
                if (writer) {
                    double *sp = a.stats + ((size_t)chain * a.n_out + orow) * BFHIP_STAT_STRIDE;
                    if (NUTS) {
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
                    } else {
                        sp[BFHIP_HS_LOGP] = h_end_logp;
                        sp[BFHIP_HS_ENERGY] = h_end_E;
                        sp[BFHIP_HS_N_INT_STEP] = (double)a.cfg.n_int_step;
                        sp[BFHIP_HS_ACCEPT_STAT] = accept_stat;
                        sp[BFHIP_HS_ACCEPTED] = (double)h_accepted;
                        sp[BFHIP_HS_STEP_SIZE] = step_now;
                        sp[BFHIP_HS_STEP_SIZE_BAR] = step_bar;
                        sp[BFHIP_HS_WARMUP] = warm ? 1. : 0.;
                        sp[BFHIP_HS_ENERGY_CHANGE] = h_de;
                        sp[BFHIP_HS_DIVERGING] = (double)diverged;
                        sp[10] = 0.;
                    }
                }
                double *xp = a.samples + ((size_t)chain * a.n_out + orow) * d;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (dbase + 4 * r < d) xp[dbase + 4 * r] = q[r];
            }
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
                    double od = q[r] - fm[r];
                    fm[r] += od / fg_n;
                    fr[r] += 1. * od * (q[r] - fm[r]);
                    od = q[r] - bm[r];
                    bm[r] += od / bg_n;
                    br[r] += 1. * od * (q[r] - bm[r]);
                }
                if ((delta + 1) % (long)a.cfg.update_window == 0) {  // metrics.py:181-184
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (dbase + 4 * r < d) { var[r] = fr[r] / fg_n; isd[r] = 1. / bf_sqrt(var[r]); }
                    store_vec(BFHIP_VEC_VAR, var);
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
            if (i_iter < a.iter_end) new_iter = true;
            else mode = M_DONE;
        }
        GTRACE2(6);
        if (bf_any(new_iter)) {
            // next iteration: metric.random, then the tree starts at (q, p) with the proposal's value and gradient; the
            // start energy needs the kinetic energy of the new momentum: it joins the next trip's exchange
            draw_momentum(new_iter);
            if (new_iter) {
                tree_reset();
                double t_k0[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) t_k0[r] = p[r] * (var[r] * p[r]);  // metrics.py:88-91
                kin0_part = sum4(t_k0);
                need_E0 = true;
            }
        }
        }  // any_end
        GTRACE2(7);
        if (err != 0) mode = M_DONE;
    }

    if (a.gcount && tid == 0) {
        bf_atomic_add_u64(a.gcount, n_trip);
        bf_atomic_add_u64(a.gcount + 1, n_trip_h);
        bf_atomic_add_u64(a.gcount + 2, n_trip_late);
        bf_atomic_add_u64(a.gcount + 3, n_trip_skip);
    }
    // ---- write the chain state back ----
    if (real) {
        store_vec(BFHIP_VEC_Q, q);
        if (writer) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.rng[(size_t)chain * 4 + k] = rs[k];
            scp[BFHIP_SC_LOG_STEP] = log_step;
            scp[BFHIP_SC_LOG_BAR] = log_bar;
            scp[BFHIP_SC_HBAR] = hbar;
            scp[BFHIP_SC_COUNT] = count;
            scp[BFHIP_SC_FG_N] = fg_n;
            scp[BFHIP_SC_BG_N] = bg_n;
            scp[BFHIP_SC_N_SAMPLES] = n_samples;
            scp[BFHIP_SC_PREV_UPDATE] = prev_upd;
            scp[BFHIP_SC_ADAPT_WINDOW] = adapt_window;
            scp[BFHIP_SC_I_ITER] = (double)i_iter;
            scp[BFHIP_SC_ERROR] = (double)err;
            if (a.n_leapfrog && nlf) bf_atomic_add_u64(a.n_leapfrog, nlf);
        }
    }
}
