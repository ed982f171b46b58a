// bfhip_sampler.hip -- fused NUTS / HMC transitions for many chains (gfx950).
//
// Work decomposition ("tile phase / chain phase"):
//   * A workgroup of 16 wavefronts owns a GROUP of 16 chains for the whole launch: 4 waves per SIMD, so the
//     latency-bound per-chain logic of one wave hides behind the other three.
//   * Chain phase (everything O(d)): ONE WAVE PER CHAIN.  Lane l holds dimensions l*E .. l*E+E-1 of every
//     state vector (E = DP/64, 1 at d <= 64).  All tree control flow is wave-uniform (and declared so, rfl()):
//     no cross-chain divergence, per-chain scalars live once per wave in scalar registers, dot products are
//     wave reductions on the matrix pipe (two v_mfma_f64_4x4x4 and two row rotations per value).
//   * Tile phase (gradient): the batched matvecs G^T = S X^T and H (X - mu)^T of the 16 chains run on
//     v_mfma_f64_16x16x4_f64; (matrix, row tile, K part) jobs are dealt over the 16 waves.  In the plain
//     instantiation every wave owns one fixed job and keeps its A operands in registers; otherwise the
//     coefficient matrices are staged once per launch in LDS as A-operand fragments.
//   * The two layouts meet in LDS: XB (B operands, written by the chain waves) and GB (matvec results).
//     Two workgroup barriers per trip, none inside the tree logic.
//   * While at most four chains of a group are still evaluating, the plain instantiation replaces the MFMA
//     tiles by per-row FMA chains that reproduce the MFMA rounding bit for bit (see the tail path below).
//
// Every chain is an independent state machine (INIT -> LEAF ... -> iteration end -> INIT ...); one loop
// trip evaluates ONE gradient for all 16 chains of the group, whatever each chain needs it for.  The
// compute_state() call that opens every iteration (base_hmc.py:70) is a leapfrog with epsilon = 0, so
// chains never wait for each other inside an iteration or across iterations.
//
// The recursion of Tree._build_subtree (samplers/nuts.py:134-178) is flattened: leaf i of a 2^depth
// subtree is merged upwards while bit `level` of i is set; completed sub-subtrees wait on a per-chain
// stack (level 0 in registers, vectors of the other levels in global scratch, scalars in LDS).  The per-chain
// logic is time-sliced into UNITS (finish an evaluation / one merge level / end of a doubling / three pieces of
// the iteration end), one unit per chain per trip, so the workgroup barrier never waits for a long bookkeeping
// path of one chain.  Random draws are consumed in the recursion's post-order, so a chain reproduces the CPU
// oracle's trajectory for the same xoshiro stream.
//
// Template instantiations: W = DP/16 (1, 2, 4, 8); NUTS / HMC; FS (compile-time feature set; 1 = PLAIN, the common
// surrogate) or generic; FULLM (full-rank metric, bfhip_metric.h); STAMPS (diagnostic phase counters).
#include <type_traits>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "bfhip_eval.h"
#include "bfhip_metric.h"
#include "bfhip_pld.h"
#include "bfhip_oob.h"

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
// wave-uniform argument -> wave-uniform result (see rfl below)
#define uexp(x) rfl(bf_exp_ni(x))
#define ulog(x) rfl(bf_log_ni(x))
#define usqrt(x) rfl(bf_sqrt_ni(x))

#include "bfhip_sampler_defs.h"

#include "bfhip_wave.h"

template <int W>
struct SamplerGeo {
    static constexpr int DP = 16 * W, NS = 4 * W;
    static constexpr int E = DP >= 64 ? DP / 64 : 1;  // state elements per lane
    static constexpr int XS = 65;                     // XB row stride (doubles)
    static constexpr int GS = DP + 1;                 // GB row stride
    static constexpr int MAT = DP * DP;
    static constexpr bool STAGE = DP <= 64;           // coefficient fragments fit in LDS
};

// Waves (= chains) per workgroup.  16 fills the 16 columns of the MFMA tiles; at d = 128 a chain's state takes two
// registers per vector and lane, and 16 waves (128 VGPRs each) spill ~180 of them: there a workgroup is 8 waves with
// 256 VGPRs each (half-empty tiles, twice as many workgroups).  The same for the full-rank metric at any d: its
// matrix-vector products stream the chain's own d x d matrix in batches of 16 columns, and at 128 VGPRs the batches were
// spilled -- 163 VGPRs of scratch -- so that every column's load was waited for on its own.
// The pipeline density has both forms: FS = 8, sixteen waves of 128 registers (~230 spilled VGPRs: its scratch traffic evicts
// part of the coefficient fragments from the XCD's L2, 17 KB of HBM reads per leapfrog step, but sixteen chains fill the
// 16-column tiles), and FS = 9, eight waves of 256 registers (no spills, 4 x 4 x 4 tiles for its eight chains).  Measured on the
// DES shape before the outputs were compressed (tools/pld_rate.py): 4096 chains 5.7 against 5.2 x 10^7 leapfrog steps/s, 1024
// chains 1.8 against 2.6 x 10^7; launch_sampler has the rule that followed once they were.
#ifndef BF_JOB_CHUNK
#define BF_JOB_CHUNK 4
#endif
// FS = 17 (config 5's shard, at most four chains per CU): FOUR waves of 512 registers, each with a chain and with two row tiles of S
// in registers for the whole launch.  In the eight-wave form every workgroup streams the 128 KB of S fragments from L2 in every
// trip -- 33 GB/s per CU in 8-byte loads, which is what the XCD's L2 gives at that width (MI355X_MICROARCH.md, indexed rows): the
// jobs were 9.3 k of the trip's 18.7 k cycles (tools/trace_sliced.py, profiles/r05b_trace_config5.log).
// (the A operand from an accumulation register: the 256 of them hold the four row tiles for the whole launch, and the compiler's own
// allocation copied every operand to a vector register first -- two v_accvgpr_read per MFMA)
// (inline assembly hides the instruction from the compiler's hazard recogniser: the s_nop 1 in front keeps dependent MFMAs of two
// interleaved chains five issue slots apart -- the compiler's own code keeps three to four -- and BF_MFMA_Q_DONE separates the last
// one from the first read of its result)
#define BF_MFMA_Q_ACC(acc, af, xv) asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc) : "a"(af), "v"(xv))
#define BF_MFMA_Q_DONE(a0, a1, a2, a3) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3))
#define BF_SAMPLER_WAVES(W, FULLM, FS) ((FS) == 17 ? 4 : (((W) == 8 || (FULLM) || (FS) == 9 || (FS) == 10) ? 8 : 16))

// PLAIN fixes the feature set of the common surrogate at compile time (linear + quadratic configs with the
// extrapolation bound; no constraint transform, no input scaling, no decay, no cubic configs): the branches
// and the state of the optional features disappear from the instantiation.
// FULLM: full-rank metric (velocity = cov p with a per-chain covariance, adapted by Welford windows and refactorised
// every update_window iterations); the diagonal metric is the default instantiation.
// FS (feature spec): 0 = everything decided at run time; otherwise linear + quadratic configs with the bound, no input
// scaling, no cubic configs, and bit 1 (value 2) = decay penalty on, bit 2 (value 4) = constraint transform on.
// FS == 1 is the PLAIN instantiation with its register-resident A operands and tail path.
__device__ inline bool g_sliced_proof_on(const SamplerArgs &a) { return a.no_bound_proof == 0; }
__device__ inline bool g_no_quad_tiles(const SamplerArgs &a) { return a.no_quad != 0; }

template <int W, bool NUTS, bool STAMPS, int FS, int FULLM>   // (FULLM: 0 diagonal metric, 1 full-rank, 2 full-rank with the next step's velocity taken ahead)
__global__ __launch_bounds__(BF_SAMPLER_WAVES(W, FULLM, FS) * 64) void bf_sampler_kernel(DevModel m, SamplerArgs a) {
    constexpr int NWV = BF_SAMPLER_WAVES(W, FULLM, FS), NTH = NWV * 64;  // waves (= chains) of a workgroup, threads
    // FS == 8: the pipeline density (bfhip_pld.h: multi-output surrogate + Gaussian likelihood + prior); transforms, input
    // scaling, bound and decay are run-time features as in FS == 0, the polynomial itself is the two contractions of phase P
    constexpr bool PLD = FS == 8 || FS == 9 || FS == 10;   // (9: eight waves of 256 registers, for launches of at most 8 chains per CU;
    constexpr bool PLDC = FS == 10;                         //  10: the same with the DES-shaped feature set fixed at compile time -- box
                                                            //  transform, input scaling, bound, no decay term)
#ifndef BF_CHAIN_UNITS_MORE
#define BF_CHAIN_UNITS_MORE 0
#endif
    constexpr bool CHAIN_UNITS = PLD || W == 8 || FULLM || (BF_CHAIN_UNITS_MORE && !STAMPS);   // (see the end of the trip loop)
    constexpr bool PLAIN = FS == 1, SPEC = FS != 0 && !PLD;
    const bool f_quad = SPEC ? true : (PLD ? false : (bool)m.has_quad), f_bound = (SPEC || PLDC) ? true : (bool)m.use_bound;
    const bool f_decay = SPEC ? (FS & 2) != 0 : (PLDC ? false : (bool)m.use_decay), f_tr = SPEC ? (FS & 4) != 0 : (PLDC ? true : (bool)m.has_transform);
    // FS == 16: linear + quadratic + cubic configs with the bound and nothing else (BASELINE config 5's surrogate), fixed at compile time
    constexpr bool CUBIC = FS == 16 || FS == 17;
    constexpr bool AREG8 = FS == 17;   // (S fragments in registers: see BF_SAMPLER_WAVES)
    const bool f_su = SPEC ? false : (PLDC ? true : (bool)m.has_su), f_cubic = CUBIC ? true : ((SPEC || PLD) ? false : (bool)m.has_cubic);
    const bool f_link = (SPEC || PLD) ? false : (bool)m.has_link;  // Gaussian likelihood of the surrogate's output (density.py:552-560)
    const int ks_rt = PLAIN ? ((W == 2 || W == 4) ? 2 : 1) : a.ks;  // K-split of the matvec jobs (sampler_ksplit)
    // PLAIN at d <= 64: there are at most 16 matvec jobs of at most 8 k-steps, so wave w runs the SAME job
    // (matrix, row tile, K part) on every trip and keeps its A operands in registers for the whole launch:
    // the coefficient matrices are not staged in LDS at all.
    constexpr bool AREG = PLAIN && W <= 4;
    constexpr bool TAIL = AREG;
    // Outside the bound's ellipsoid a linear + quadratic surrogate needs no second pass: S x_0 of the projected point follows from
    // S x by linearity (bfhip_oob.h).  Only cubic configs evaluate again, in a trip of their own (mode M_OOB).
    constexpr int KS_P = (W == 2 || W == 4) ? 2 : 1, KPJ_P = (4 * W) / KS_P, NJOB_P = 2 * W * KS_P;
    using G = SamplerGeo<W>;
    constexpr int DP = G::DP, NS = G::NS, E = G::E, XS = G::XS, GS = G::GS, MAT = G::MAT;
    constexpr int MAXL = BFHIP_MAX_TREEDEPTH;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *XB = lds;                         // [3][NS][XS]   B operands
    double *LS = XB + 3 * NS * XS;            // [16][MAXL][LS_N] per-chain stack scalars
    int *alive = (int *)(LS + 16 * MAXL * LS_N);  // [2] any chain not done | [2] mask of the chains evaluating | [2] some chain
                                                  // is not proven inside the bound (all by trip parity)
    double *CS = LS + 16 * MAXL * LS_N + 4;   // (an even offset: the regions behind keep their 16-byte alignment)   // [16][CS_N]    cold per-chain scalars (kept out of the VGPR budget)
    double *PDL = CS + 16 * CS_N;             // [PD_N][DP]    per-dimension table (rarely used rows are read from here)
    double *FR = PDL + PD_N * DP;             // staged A fragments: S | H | H_decay
    // the matvec results [gbn][16][GS] (slot = enabled matrix x K part) come last: every region above keeps a
    // compile-time LDS offset
    // plain kernel: row-major padded copies of S and H (RM) and the chains' x in plain layout (XP) feed the VALU
    // matvec that replaces the MFMA jobs while only a few chains of the group are still evaluating
    constexpr int RS = DP + 2;                // RM row stride (16-byte aligned rows, conflict-free b128 reads)
    double *RM = FR;                          // [2][DP][RS]
    double *XP = RM + 2 * DP * RS;            // [2][16][DP]
    double *GB = AREG ? XP + 2 * 16 * DP
                      : FR + (G::STAGE ? MAT * ((f_quad ? 1 : 0) + (f_bound ? 1 : 0) + (f_decay ? 1 : 0)) : 0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index == chain index in the group
    const int cpg = a.cpg > 0 ? a.cpg : NWV;  // chains of this workgroup (the other waves only run matvec jobs)
    const int chain = blockIdx.x * cpg + w;
    const bool real = w < cpg && chain < a.n_chain;
    const int d = m.d;

    // ---- stage the coefficient matrices (A-operand fragments) in LDS ----
    const double *Sf = m.Sf, *Hf = m.Hf, *Hdf = m.Hdf;
    if constexpr (G::STAGE && !AREG) {
        double *pS = FR, *pH = pS + (f_quad ? MAT : 0), *pD = pH + (f_bound ? MAT : 0);
        if (f_quad)
            for (int i = tid; i < MAT / 2; i += NTH) ((d2_t *)pS)[i] = ((const d2_t *)m.Sf)[i];
        if (f_bound)
            for (int i = tid; i < MAT / 2; i += NTH) ((d2_t *)pH)[i] = ((const d2_t *)m.Hf)[i];
        if (f_decay)
            for (int i = tid; i < MAT / 2; i += NTH) ((d2_t *)pD)[i] = ((const d2_t *)m.Hdf)[i];
        Sf = pS; Hf = pH; Hdf = pD;
    }
    double afr[AREG ? KPJ_P : 1];
    if constexpr (AREG) {
        const int slot_m = w / (W * KS_P), rem = w % (W * KS_P), t = rem / KS_P, kp = rem % KS_P;
        const double *Af = (slot_m == 0 ? m.Sf : m.Hf) + (t * NS + kp * KPJ_P) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KPJ_P; ++s) afr[s] = (w < NJOB_P) ? Af[s * 64] : 0.;
    }
    double afr8[AREG8 ? 4 : 1][AREG8 ? NS : 1];   // row tiles w and w + 4 of S, then of H, all k-steps
    if constexpr (AREG8) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                afr8[jt][s] = m.Sf[((w + 4 * jt) * NS + s) * 64 + lane];
                afr8[2 + jt][s] = m.Hf[((w + 4 * jt) * NS + s) * 64 + lane];
            }
    }
    if constexpr (TAIL) {
        // frag[(t * NS + s) * 64 + l] = M[16 t + (l & 15)][4 s + (l >> 4)]  ->  RM[b][row][col]
        for (int i = tid; i < MAT; i += NTH) {
            const int l = i & 63, s = (i >> 6) % NS, t = (i >> 6) / NS;
            const int row = 16 * t + (l & 15), col = 4 * s + (l >> 4);
            RM[(0 * DP + row) * RS + col] = m.Sf[i];
            RM[(1 * DP + row) * RS + col] = m.Hf[i];
        }
    }
    if (tid < 8) alive[tid] = 0;
    for (int i = tid; i < PD_N * DP; i += NTH) PDL[i] = m.pd[i];
    // cubic configs: the coefficient tables behind the matvec results when they fit (sampler_cubic_lds: config 5's 16
    // masked inputs are 36 KB), the masks of this lane in registers
    const bool cub_l = f_cubic && a.cub_lds != 0;
    double *CUB = GB + (((size_t)a.gbn * 16 * GS + 1) & ~(size_t)1);  // [n2 n2] A2t | [n2 n2] A2 | [nc3 n3 n3 16] T3x
    // pipeline density: its regions behind the matvec results (there are no cubic tables then)
    PldLds PL;
    if constexpr (PLD) {
        PL = pld_lds(CUB, DP, m.pld, NWV == 8 ? 8 : 16, NWV == 8 && a.pld_cl != 0);   // (the eight-chain forms: compact B-operand rows)
        pld_stage(m.pld, PL, DP, tid, NTH);
    }
    int mk2 = 0, mk3 = 0, pj2[E], pj3[E];
#pragma unroll
    for (int e = 0; e < E; ++e) pj2[e] = pj3[e] = -1;
    if (cub_l) {
        const int n22 = m.n2 * m.n2;
        for (int i = tid; i < n22; i += NTH) { CUB[i] = m.A2t[i]; CUB[n22 + i] = m.A2[i]; }
        // T3t is [k][l][j]; in LDS it is [k][l / 2][j][l & 1] with l padded to a multiple of 16 (zero where l >= n3): lane (j, kq)
        // of the contraction reads T[j, k, l], T[j, k, l + 1] as one 16-byte pair, sixteen lanes 256 bytes in a row
        const int n3 = m.n3, nc3 = (n3 + 15) >> 4;
        double *T3x = lds + (((size_t)(CUB + 2 * n22 - lds) + 1) & ~(size_t)1);
        for (int i = tid; i < nc3 * n3 * n3 * 16; i += NTH) {
            const int jj = (i >> 1) % n3, l = 2 * (((i >> 1) / n3) % (8 * nc3)) + (i & 1), k = (i >> 1) / (n3 * 8 * nc3);
            T3x[i] = l < n3 ? m.T3t[((size_t)k * n3 + l) * n3 + jj] : 0.;
        }
        mk2 = lane < m.n2 ? m.mask2[lane] : 0;
        mk3 = lane < m.n3 ? m.mask3[lane] : 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = lane * E + e;
            pj2[e] = (dim < m.DP && m.n2 > 0) ? m.pos2[dim] : -1;
            pj3[e] = (dim < m.DP && m.n3 > 0) ? m.pos3[dim] : -1;
        }
    }

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
    // hot tree scalars in registers: largest |dE| of the tree, the weight offset, and the waiting level-0 leaf's weight
    // and accept sum (mirrors of lsw[LS_LS], lsw[LS_ACC] of level 0, which stay the backing store)
    double max_de = 0., w_off = 0., L0_W = 0., L0_acc = 0.;
    int unit = U_DONE, lev = 0, h_accepted = 0;
    double *csw = CS + w * CS_N;
    auto cs_set = [&](int i, double v) { if (lane == 0) csw[i] = v; };
    auto cs_get = [&](int i) -> double { return rfl(csw[i]); };
    unsigned long long nlf = 0;
    const bool lane_ok = lane * E < DP;  // lanes beyond the padded dimension idle (DP < 64)
    double *sbase = a.scratch + ((size_t)(real ? chain : 0) * a.nslot) * DP + lane * E;
    double *scp = a.sc + (size_t)(real ? chain : 0) * BFHIP_SC_N;
    double *vecp = a.vec + (size_t)(real ? chain : 0) * BFHIP_VEC_N * d;
    double *lsw = LS + w * (MAXL * LS_N);
    const int nw = a.cfg.n_warmup;
    // full-rank metric: this chain's matrices (bfhip_metric.h) and the velocity of a momentum
    const size_t msz = (size_t)d * d;
    double *matp = FULLM ? a.mat + (size_t)(real ? chain : 0) * BF_MAT_N * msz : nullptr;
    double vcur[FULLM ? E : 1], L0v[FULLM ? E : 1];  // velocities of the current momentum and of the waiting level-0 leaf
    auto velocity = [&](const double (&pv)[E], double (&out)[E]) {
        if constexpr (FULLM) {
            bf_velocity_full<E>(matp + BF_MAT_COV * msz, pv, out, d, lane);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) out[e] = var[e] * pv[e];
        }
    };

    // Full-rank metric: the velocity of the NEXT step's half-step momentum, taken in the pass that takes this step's final velocity
    // (bf_velocity_full2).  p_ahead is the momentum it belongs to: the next phase A uses the cached product only if its own
    // half-step momentum has exactly these bits (the tree goes on from this leaf with the same step; at a turn of the direction,
    // a new tree or a step-size change it does not, and the pass is taken as before) -- the same numbers either way.
    double p_ahead[FULLM ? E : 1], v_ahead[FULLM ? E : 1];
    bool ahead_ok = false;
    constexpr bool VEL_AHEAD = FULLM == 2;

    auto velocity3 = [&](const double (&a0)[E], const double (&a1)[E], const double (&a2)[E], double (&o0)[E], double (&o1)[E], double (&o2)[E]) {
        if constexpr (FULLM) {
            bf_velocity_full3<E>(matp + BF_MAT_COV * msz, a0, a1, a2, o0, o1, o2, d, lane);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) { o0[e] = var[e] * a0[e]; o1[e] = var[e] * a1[e]; o2[e] = var[e] * a2[e]; }
        }
    };

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
            if constexpr (FULLM) p[e] = (dim < d) ? z : 0.;
            else p[e] = (dim < d) ? (1. / sqrt(var[e])) * z : 0.;
            g[e] = 0.;
        }
        if constexpr (FULLM) bf_solve_lt<E>(matp + BF_MAT_CHOL_ROWS * msz, p, d, lane);  // metrics.py:123-127
    };
#pragma unroll
    for (int e = 0; e < E; ++e) { q[e] = 0.; p[e] = 0.; g[e] = 0.; var[e] = 1.; TLp[e] = 0.; TPs[e] = 0.; TPq[e] = 0.; PF0[e] = 0.; PF1[e] = 0.; PF2[e] = 0.; PF3[e] = 0.; L0p[e] = 0.; L0q[e] = 0.; }
    if (real) {
        for (int k = 0; k < 4; ++k) rs[k] = rfl((uint64_t)a.rng[(size_t)chain * 4 + k]);
        cs_set(CS_LOG_STEP, rfl(scp[BFHIP_SC_LOG_STEP]));
        cs_set(CS_LOG_BAR, rfl(scp[BFHIP_SC_LOG_BAR]));
        cs_set(CS_HBAR, rfl(scp[BFHIP_SC_HBAR]));
        cs_set(CS_SMU, rfl(scp[BFHIP_SC_MU]));
        cs_set(CS_COUNT, rfl(scp[BFHIP_SC_COUNT]));
        cs_set(CS_STEP_NOW, uexp(rfl(scp[BFHIP_SC_LOG_STEP])));  // exp(log_step), exp(log_bar): what the statistics report
        cs_set(CS_STEP_BAR, uexp(rfl(scp[BFHIP_SC_LOG_BAR])));
        i_iter = rfl((int)scp[BFHIP_SC_I_ITER]);
        err = rfl((int)scp[BFHIP_SC_ERROR]);
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
    // BF_TRACE=<n> (tuning builds only): wave 0 of workgroup 0 records s_memtime at up to 16 points of its first n
    // trips in LDS and dumps them to the stamps buffer
    int trip_no = 0;
    (void)trip_no;
#ifdef BF_TRACE
    __shared__ unsigned long long TRC[BF_TRACE * 16];
#define TRACE(k) do { if ((!PLD || (k) <= 10) && w == 0 && blockIdx.x == 0 && trip_no < BF_TRACE && lane == 0) TRC[trip_no * 16 + (k)] = clock64(); } while (0)
#ifdef BF_TRACE_CUBIC   // (stamps 11-13 inside cubic_lds instead of the leaf unit's)
#define TRACEU(k) do { } while (0)
#define TRACEC(k) TRACE(k)
#else
#define TRACEU(k) TRACE(k)
#define TRACEC(k) do { } while (0)
#endif
#define TRACEP(k) do { if (PLD && w == 0 && blockIdx.x == 0 && trip_no < BF_TRACE && lane == 0) TRC[trip_no * 16 + (k)] = clock64(); } while (0)
    for (int i = threadIdx.x; i < BF_TRACE * 16; i += NTH) TRC[i] = 0;
#else
#define TRACE(k) do { } while (0)
#define TRACEP(k) do { } while (0)
#define TRACEU(k) do { } while (0)
#define TRACEC(k) do { } while (0)
#endif

    auto run_unit = [&](bool have_ev, double E_new, double logp_new) {
        // ================= per-chain state machine: ONE unit of work per trip =================
        // (wave-uniform control flow; the barrier-to-barrier critical path is the longest single unit.)
        // (Running the merge levels / the doubling end that follow a leaf in the leaf's own trip was measured slower
        // twice: with the subtree stack in global memory a fused merge consumes its loads at once, and the larger live
        // ranges triple the register spills; with one unit per trip the operands are prefetched a trip ahead.)
        if (unit == U_EVAL) {
            if (__builtin_expect(have_ev && mode == M_INIT, 0)) {
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
                    w_off = 0.;
                    max_de = 0.;
                    depth = 0; acc_sum = 0.; n_prop = 0; diverged = 0; i_leaf = 0;
                    eps = uexp(i_iter < nw ? cs_get(CS_LOG_STEP) : cs_get(CS_LOG_BAR));  // step_size.py:25-29
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
                    if (fabs(dE) > fabs(max_de)) max_de = dE;
                    TRACEU(11);
                    cs_set(CS_T_E, E_new);
                    cs_set(CS_T_LOGP, logp_new);
                    T_acc = 0.; lev = 0;
                    if (fabs(dE) < a.cfg.max_change) {
                        // multinomial weight exp(log_size) = exp(-dE), kept in the linear domain relative to
                        // a running offset w_off (exact streaming log-sum-exp; rescales are rare)
                        double aw = -dE - w_off;
                        if (aw > 600.) {
                            const double sc_ = uexp(-aw);
                            cs_set(CS_TREE_W, cs_get(CS_TREE_W) * sc_);
                            if (lane == 0)
                                for (int l2 = 0; l2 < depth; ++l2) lsw[l2 * LS_N + LS_LS] *= sc_;
                            L0_W *= sc_;
                            w_off = w_off + aw;
                            aw = 0.;
                        }
                        // (inlined exps were tried here -- a degree-13 Estrin polynomial, then a 64-entry table with a
                        // degree-5 polynomial, 2 ulp: both lose 4-11 % to the extra live registers of this kernel at its
                        // 128-VGPR budget; the out-of-line libm call stays)
                        T_W = uexp(aw);
                        const double pacc = (w_off == 0.) ? T_W : uexp(-dE);
                        T_acc = pacc > 1. ? 1. : pacc;
                        TRACEU(12);
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
                                d0 += ps0 * (FULLM ? L0v[FULLM ? e : 0] : var[e] * L0p[e]);  // nuts.py:150-151
                                d1 += ps0 * (FULLM ? vcur[FULLM ? e : 0] : var[e] * p[e]);
                            }
                            TRACEU(13);
                            { double r2[2] = {d0, d1}; wave_sum_n<2>(r2); d0 = r2[0]; d1 = r2[1]; }
                            TRACEU(14);
                            T_acc = L0_acc + T_acc;  // :173
                            const double Wsum = L0_W + T_W;
                            if (Wsum != Wsum) err = 2;
                            const double u = bf_u01(bf_xoshiro_next(rs));  // :163-167, drawn even when turning
                            TRACEU(15);
                            if ((d0 <= 0.) || (d1 <= 0.)) {
                                unit = U_ABORT;
                                lev = 1;
                            } else {
                                if (!((u * Wsum < T_W) || (u == 0.))) {
#pragma unroll
                                    for (int e = 0; e < E; ++e) TPq[e] = L0q[e];
                                    cs_set(CS_T_E, rfl(lsw[LS_E]));
                                    cs_set(CS_T_LOGP, rfl(lsw[LS_LOGP]));
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
                        double h_accept_stat = uexp(h_dE);
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
            double vAa[E], vBa[E], vCa[E];
            velocity3(A, B, TLp, vAa, vBa, vCa);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                psum[e] = S1[e] + TPs[e];
                const double vA = vAa[e], vB = vBa[e], vC = vCa[e], vD = FULLM ? vcur[FULLM ? e : 0] : var[e] * p[e];
                d0 += psum[e] * vA;  // nuts.py:150-151
                d1 += psum[e] * vD;
                const double ps1 = S1[e] + TLp[e];  // :155-157
                d2 += ps1 * vA;
                d3 += ps1 * vC;
                const double ps2 = B[e] + TPs[e];   // :158-160
                d4 += ps2 * vB;
                d5 += ps2 * vD;
            }
            {   // this unit only runs for lev >= 1 (level-0 merges are inline in the leaf), so all six checks apply
                double r6[6] = {d0, d1, d2, d3, d4, d5};
                wave_sum_n<6>(r6);
                d0 = r6[0]; d1 = r6[1]; d2 = r6[2]; d3 = r6[3]; d4 = r6[4]; d5 = r6[5];
            }
            bool turning = (d0 <= 0.) || (d1 <= 0.);
            if (lev >= 1) turning = turning || (d2 <= 0.) || (d3 <= 0.) || (d4 <= 0.) || (d5 <= 0.);
            const double *lsp = lsw + lev * LS_N;
            T_acc = rfl(lsp[LS_ACC]) + T_acc;  // :173
            // nuts.py:163-167 run even when THIS merge's check says turning: the draw is consumed.
            // logbern(ls2 - logaddexp(ls1, ls2))  <=>  U * (W1 + W2) < W2
            const double Wsum = rfl(lsp[LS_LS]) + T_W;
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
                    cs_set(CS_T_E, rfl(lsp[LS_E]));
                    cs_set(CS_T_LOGP, rfl(lsp[LS_LOGP]));
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
                const double tree_W = cs_get(CS_TREE_W);
                if (T_W != T_W || tree_W != tree_W) err = 2;
                const double u = bf_u01(bf_xoshiro_next(rs));
                if ((u * tree_W < T_W) || (u == 0.)) {
                    stv(SL_PROP_Q, TPq);
                    cs_set(CS_PROP_E, cs_get(CS_T_E));
                    cs_set(CS_PROP_LOGP, cs_get(CS_T_LOGP));
                }
                cs_set(CS_TREE_W, tree_W + T_W);  // :85
            }
            double d0 = 0., d1 = 0., d2 = 0., d3 = 0., d4 = 0., d5 = 0.;
            double vTa[E], vLa[E], vRa[E];
            velocity3(TLp, oldL, oldR, vTa, vLa, vRa);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ps[e] += TPs[e];  // :86 (in place)
                const double vN = FULLM ? vcur[FULLM ? e : 0] : var[e] * p[e], vT = vTa[e], vL = vLa[e], vR = vRa[e];
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
            {
                double r6[6] = {d0, d1, d2, d3, d4, d5};
                wave_sum_n<6>(r6);
                d0 = r6[0]; d1 = r6[1]; d2 = r6[2]; d3 = r6[3]; d4 = r6[4]; d5 = r6[5];
            }
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
            const double accept_stat = NUTS ? acc_sum / (double)n_prop : cs_get(CS_HACC);  // nuts.py:186
            double log_step = cs_get(CS_LOG_STEP), log_bar = cs_get(CS_LOG_BAR);
            if (warm && a.cfg.adapt_step_size) {  // step_size.py:31-45
                const double count = cs_get(CS_COUNT);
                const double wgt = 1. / (count + a.cfg.t_0);
                const double hbar = ((1. - wgt) * cs_get(CS_HBAR) + wgt * (a.cfg.target_accept - accept_stat));
                log_step = cs_get(CS_SMU) - hbar * usqrt(count) / a.cfg.gamma;
                const double mk = uexp(-a.cfg.k * ulog(count));  // count ** -k
                log_bar = mk * log_step + (1. - mk) * log_bar;
                cs_set(CS_HBAR, hbar);
                cs_set(CS_LOG_STEP, log_step);
                cs_set(CS_LOG_BAR, log_bar);
                cs_set(CS_COUNT, count + 1.);
                cs_set(CS_STEP_NOW, uexp(log_step));
                cs_set(CS_STEP_BAR, uexp(log_bar));
            }
            const double prop_E = cs_get(CS_PROP_E), prop_logp = cs_get(CS_PROP_LOGP);
            const int orow = i_iter - a.iter_out0;
            if (orow >= 0 && orow < a.n_out && lane == 0) {
                double *st = a.stats + ((size_t)chain * a.n_out + orow) * BFHIP_STAT_STRIDE;
                if (NUTS) {
                    st[BFHIP_NS_LOGP] = prop_logp;
                    st[BFHIP_NS_ENERGY] = prop_E;
                    st[BFHIP_NS_TREE_DEPTH] = (double)depth;
                    st[BFHIP_NS_TREE_SIZE] = (double)n_prop;
                    st[BFHIP_NS_MEAN_TREE_ACCEPT] = accept_stat;
                    st[BFHIP_NS_STEP_SIZE] = cs_get(CS_STEP_NOW);
                    st[BFHIP_NS_STEP_SIZE_BAR] = cs_get(CS_STEP_BAR);
                    st[BFHIP_NS_WARMUP] = warm ? 1. : 0.;
                    st[BFHIP_NS_ENERGY_CHANGE] = prop_E - start_energy;
                    st[BFHIP_NS_MAX_ENERGY_CHANGE] = max_de;
                    st[BFHIP_NS_DIVERGING] = (double)diverged;
                } else {
                    st[BFHIP_HS_LOGP] = prop_logp;
                    st[BFHIP_HS_ENERGY] = prop_E;
                    st[BFHIP_HS_N_INT_STEP] = (double)a.cfg.n_int_step;
                    st[BFHIP_HS_ACCEPT_STAT] = accept_stat;
                    st[BFHIP_HS_ACCEPTED] = (double)h_accepted;
                    st[BFHIP_HS_STEP_SIZE] = cs_get(CS_STEP_NOW);
                    st[BFHIP_HS_STEP_SIZE_BAR] = cs_get(CS_STEP_BAR);
                    st[BFHIP_HS_WARMUP] = warm ? 1. : 0.;
                    st[BFHIP_HS_ENERGY_CHANGE] = cs_get(CS_HDE);
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
            if (FULLM && warm && a.cfg.adapt_metric) {
                // QuadMetricFullAdapt.update: metrics.py:294-324, _WeightedCovariance.add_sample :401-407
                double fg_n = rfl(scp[BFHIP_SC_FG_N]), bg_n = rfl(scp[BFHIP_SC_BG_N]);
                double n_samples = rfl(scp[BFHIP_SC_N_SAMPLES]), prev_upd = rfl(scp[BFHIP_SC_PREV_UPDATE]);
                double adapt_window = rfl(scp[BFHIP_SC_ADAPT_WINDOW]);
                const long delta = (long)(n_samples - prev_upd);
                double fm[E], bm[E], od[E], nd[E];
                load_vec(BFHIP_VEC_FG_MEAN, fm, 0.);
                load_vec(BFHIP_VEC_BG_MEAN, bm, 0.);
                double *fgT = matp + BF_MAT_FG * msz, *bgT = matp + BF_MAT_BG * msz, *covT = matp + BF_MAT_COV * msz;
                fg_n += 1.;
#pragma unroll
                for (int e = 0; e < E; ++e) { od[e] = q[e] - fm[e]; fm[e] += od[e] / fg_n; nd[e] = q[e] - fm[e]; }
                const bool refresh = (delta + 1) % (long)a.cfg.update_window == 0;   // _update_from_weightvar: :287-292
                bf_welford_cov<E>(fgT, nd, od, d, lane, refresh ? covT : nullptr, fg_n);
                bg_n += 1.;
#pragma unroll
                for (int e = 0; e < E; ++e) { od[e] = q[e] - bm[e]; bm[e] += od[e] / bg_n; nd[e] = q[e] - bm[e]; }
                bf_welford_cov<E>(bgT, nd, od, d, lane);
                if (refresh) {  // (covT = fgT / fg_n was written by the foreground update above)
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
                    fg_n = bg_n;
                    bg_n = 10.;
                    prev_upd = n_samples;
                    if (a.cfg.doubling) adapt_window *= 2.;
                }
                n_samples += 1.;
                store_vec(BFHIP_VEC_FG_MEAN, fm);
                store_vec(BFHIP_VEC_BG_MEAN, bm);
                if (lane == 0) {
                    scp[BFHIP_SC_FG_N] = fg_n;
                    scp[BFHIP_SC_BG_N] = bg_n;
                    scp[BFHIP_SC_N_SAMPLES] = n_samples;
                    scp[BFHIP_SC_PREV_UPDATE] = prev_upd;
                    scp[BFHIP_SC_ADAPT_WINDOW] = adapt_window;
                }
            }
            // QuadMetricDiagAdapt.update: metrics.py:186-211, _WeightedVariance.add_sample :354-360
            if (!FULLM && warm && a.cfg.adapt_metric) {
                double fg_n = rfl(scp[BFHIP_SC_FG_N]), bg_n = rfl(scp[BFHIP_SC_BG_N]);
                double n_samples = rfl(scp[BFHIP_SC_N_SAMPLES]), prev_upd = rfl(scp[BFHIP_SC_PREV_UPDATE]);
                double adapt_window = rfl(scp[BFHIP_SC_ADAPT_WINDOW]);
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
        if (__builtin_expect(unit == U_ABORT, 0)) {
            // unwind: every pending ancestor adds its left half's accept_sum (nuts.py:173)
            for (int al = (diverged ? 0 : lev); al < depth; ++al)
                if ((i_leaf >> al) & 1) T_acc = rfl(lsw[al * LS_N + LS_ACC]) + T_acc;
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
                    if constexpr (FULLM) {
#pragma unroll
                        for (int e = 0; e < E; ++e) L0v[e] = vcur[e];
                    }
                } else {
                    const int slot = SL_STACK + 4 * lev;
                    stv(slot + 0, TLp); stv(slot + 1, p); stv(slot + 2, TPs); stv(slot + 3, TPq);
                }
                if (lane == 0) {
                    double *lsp = lsw + lev * LS_N;
                    // (a level-0 subtree is the fresh leaf itself: its energy and logp are this trip's evaluation,
                    // no need to read back what the leaf just parked in LDS)
                    lsp[LS_LS] = T_W; lsp[LS_ACC] = T_acc;
                    lsp[LS_E] = (lev == 0 && have_ev) ? E_new : cs_get(CS_T_E);
                    lsp[LS_LOGP] = (lev == 0 && have_ev) ? logp_new : cs_get(CS_T_LOGP);
                }
                if (lev == 0) { L0_W = T_W; L0_acc = T_acc; }
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

    // enabled coefficient matrices in slot order: mat0 = id of the first, mat1 = id of the second (the third is 2)
    const int n_mat = (f_quad ? 1 : 0) + (f_bound ? 1 : 0) + (f_decay ? 1 : 0);
    const int mat0 = f_quad ? 0 : (f_bound ? 1 : 2);
    const int mat1 = f_quad ? (f_bound ? 1 : 2) : 2;

    // result of enabled matrix slot_m for this chain: the K parts are added in a fixed order
    const int slot_S = 0, slot_H = f_quad ? 1 : 0, slot_D = slot_H + (f_bound ? 1 : 0);
    auto gb_read = [&](int slot_m, int dim) -> double {
        const double *gp = GB + ((slot_m * ks_rt) * 16 + w) * GS + dim;
        double r = gp[0];
        if (ks_rt > 1) r += gp[16 * GS];
        if (ks_rt > 2) { r += gp[2 * 16 * GS]; r += gp[3 * 16 * GS]; }
        return r;
    };

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

    // The cubic configs of this wave's chain with the tables in LDS (cub_l): the contributions to the gradient entries this lane
    // owns (gc2: cubic-2, gc3: cubic-3; a dimension gets at most one of each) and the lane's part of the value.  xe: the
    // evaluation point, element e of lane l is dimension l E + e.  Called in phase C -- or, when the workgroup has at least as
    // many chain-less waves as chains (wave_layout_cpg: config 5's shard, 4 chains on 8 waves), right after barrier B1 by the
    // chain's wave while the chain-less waves take ALL the matvec jobs: the contraction (a third of the trip) then runs beside
    // the jobs, which wait for their A operands from L2 most of the time, instead of behind them.
    constexpr bool XT = W == 8 && !FULLM;
    auto cubic_lds = [&](const double (&xe)[E], double (&gc2)[E], double (&gc3)[E]) -> double {
        const int jl = lane & 15, kq = lane >> 4;
        auto xl_ = [&](int dim) {
            const double a0 = __shfl(xe[0], dim / E, 64);
            if constexpr (E > 1) { const double a1 = __shfl(xe[E - 1], dim / E, 64); return (dim % E) ? a1 : a0; }
            return a0;
        };
        const double xm2 = xl_(mk2), xm3 = xl_(mk3);
        double fsum = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) gc2[e] = gc3[e] = 0.;
        auto fetch_l = [&](const int (&pj)[E], int jb, double val, double (&dst)[E]) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool mine = pj[e] >= jb && pj[e] < jb + 16;
                const double gv = __shfl(val, mine ? pj[e] - jb : 0, 64);
                if (mine) dst[e] += gv;
            }
        };
        TRACEC(11);
        const int n2 = m.n2, n3 = m.n3;
        const double *A2t_l = CUB, *A2_l = CUB + n2 * n2;
        const double *T3_l = lds + (((size_t)(CUB + 2 * n2 * n2 - lds) + 1) & ~(size_t)1);
        const int nc3 = (n3 + 15) >> 4;
        if (AREG8 && n2 == 16 && n3 == 16 && !a.cub_loops) {   // (the four-wave form: where the contraction is on the trip's critical path)
            // Sixteen masked inputs in both configs (config 5): the loops below written out as ONE basic block -- the same sums in the
            // same order, but the cubic-2 part's broadcasts, table reads and lane exchanges (1.4 k cycles of round trips on their own)
            // travel under the cubic-3 part's fma chains instead of in front of them.
            double a1[4], a2[4], xk2[4], xk3[4], mk[4] = {0., 0., 0., 0.};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * u + kq;
                xk2[u] = __shfl(xm2, k, 64);
                xk3[u] = __shfl(xm3, k, 64);
                a1[u] = A2t_l[k * 16 + jl];
                a2[u] = A2_l[k * 16 + jl];
            }
            const double xj2 = __shfl(xm2, jl, 64), xj3 = __shfl(xm3, jl, 64);
            const double *Tb = T3_l + ((size_t)kq * 8 * 16 + jl) * 2;   // k = 4 u + kq: stride 4 * 8 * 16 * 2 doubles per u
#pragma unroll
            for (int l2 = 0; l2 < 8; l2 += 2) {
                d2_t ta[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int u = 0; u < 4; ++u) ta[i][u] = *(const d2_t *)(Tb + (size_t)u * (4 * 8 * 16 * 2) + (l2 + i) * 16 * 2);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const double x0 = readlane_f64(xm3, 2 * (l2 + i)), x1 = readlane_f64(xm3, 2 * (l2 + i) + 1);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        mk[u] = __builtin_fma(ta[i][u][0], x0, mk[u]);
                        mk[u] = __builtin_fma(ta[i][u][1], x1, mk[u]);
                    }
                }
            }
            double v1 = 0., v2 = 0., sacc = 0.;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v1 += a1[u] * xk2[u];
                v2 += a2[u] * (xk2[u] * xk2[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sacc = __builtin_fma(mk[u], xk3[u], sacc);
            v1 = swap32_add_f64(swap16_add_f64(v1));
            v2 = swap32_add_f64(swap16_add_f64(v2));
            sacc = swap32_add_f64(swap16_add_f64(sacc));
            const double gj2 = 2. * xj2 * v1 + v2;
            if (kq == 0) fsum += xj2 * xj2 * v1;
            if (kq == 0) fsum += xj3 * (0.5 * sacc) * (1. / 3.);
            fetch_l(pj2, 0, gj2, gc2);
            fetch_l(pj3, 0, 0.5 * sacc, gc3);
            TRACEC(12);
            TRACEC(13);
            return fsum;
        }
        // (The loads of a batch are issued together, then the sums run in the original order: with one LDS
        // round trip per term the contraction was 13 k of config 5's 29 k cycles per trip,
        // tools/trace_sliced.py.  Terms past the tables' ends are +0 and leave the sums as they were.)
        for (int jb = 0; jb < n2; jb += 16) {
            const int j = jb + jl;
            const bool on = j < n2;
            const int jc = on ? j : 0;
            const double xj = __shfl(xm2, jc, 64);   // (on its way before the sums, used after them)
            double v1 = 0., v2 = 0.;
            for (int kk = 0; kk < n2; kk += 16) {
                double a1[4], a2[4], xk[4];
                if (jb + 16 <= n2 && kk + 16 <= n2) {   // (wave-uniform: a full tile needs no guards)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = kk + 4 * u + kq;
                        xk[u] = __shfl(xm2, k, 64);
                        a1[u] = A2t_l[k * n2 + j];
                        a2[u] = A2_l[k * n2 + j];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = kk + 4 * u + kq;
                        const bool ok = on && k < n2;
                        const int kc = ok ? k : 0;
                        xk[u] = __shfl(xm2, kc, 64);
                        const double t1 = A2t_l[kc * n2 + jc], t2 = A2_l[kc * n2 + jc];
                        a1[u] = ok ? t1 : 0.;
                        a2[u] = ok ? t2 : 0.;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v1 += a1[u] * xk[u];
                    v2 += a2[u] * (xk[u] * xk[u]);
                }
            }
            v1 = swap32_add_f64(swap16_add_f64(v1));
            v2 = swap32_add_f64(swap16_add_f64(v2));
            const double gj2 = 2. * xj * v1 + v2;
            if (on && kq == 0) fsum += xj * xj * v1;
            fetch_l(pj2, jb, gj2, gc2);
        }
        TRACEC(12);
        // cubic-3, lane (j, kq): sum_k x_k M[j, k] with M[j, k] = sum_l T[j, k, l] x_l taken as ONE fma chain over l from zero -- what a
        // matrix instruction computes for a tile of M (and sixteen independent chains per lane at 16 inputs, where the (j, lq) form
        // before it had one chain over k: it was the longest piece of config 5's trip, tools/trace_sliced.py).  The lane's k are
        // 16 g + 4 u + kq; x_l is a scalar operand, x_k the only broadcast per lane.
        for (int jb = 0; jb < n3; jb += 16) {
            const int j = jb + jl;
            const bool on = j < n3;
            const int jc = on ? j : 0;
            const int nl2 = 8 * nc3;
            const double xj = __shfl(xm3, jc, 64);
            double sacc = 0.;
            for (int g = 0; g < nc3; ++g) {
                double mk[4] = {0., 0., 0., 0.};
                int Tk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 16 * g + 4 * u + kq;
                    Tk[u] = ((k < n3 ? k : 0) * nl2 * n3 + jc) * 2;
                }
                for (int l2 = 0; l2 < nl2; l2 += 2) {
                    d2_t ta[2][4];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int u = 0; u < 4; ++u) ta[i][u] = *(const d2_t *)(T3_l + Tk[u] + (l2 + i) * n3 * 2);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int l = 2 * (l2 + i);   // (uniform; past n3 the table is zero and any finite x will do)
                        const double x0 = readlane_f64(xm3, l < n3 ? l : 0), x1 = readlane_f64(xm3, l + 1 < n3 ? l + 1 : 0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            mk[u] = __builtin_fma(ta[i][u][0], x0, mk[u]);
                            mk[u] = __builtin_fma(ta[i][u][1], x1, mk[u]);
                        }
                    }
                }
                double xk[4];   // (the four broadcasts on their way together, then the chain)
#pragma unroll
                for (int u = 0; u < 4; ++u) xk[u] = __shfl(xm3, 16 * g + 4 * u + kq < n3 ? 16 * g + 4 * u + kq : 0, 64);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) sacc = __builtin_fma(mk[u], 16 * g + 4 * u + kq < n3 ? xk[u] : 0., sacc);
            }
            sacc = swap32_add_f64(swap16_add_f64(sacc));
            if (on && kq == 0) fsum += xj * (0.5 * sacc) * (1. / 3.);
            fetch_l(pj3, jb, 0.5 * sacc, gc3);
        }
        TRACEC(13);
        return fsum;
    };
    // (d = 128 only, where they were measured: the 128-register instantiations of d <= 64 and the full-rank one have no
    // register to spare -- with these paths compiled in they spilled 25 % more VGPRs, the full-rank one 135 instead of 90)
    const bool cub_early = XT && !AREG8 && cub_l && 2 * cpg <= NWV;   // (fixed for the launch, the same in every wave)
    for (int trip = 0;; ++trip) {
        trip_no = trip;
        TRACE(0);
        // ================= phase A: first half of the leapfrog, B operands =================
        double xs[E], jac[E], gj[E], xo[E], xea[E];
        double logdet = 0.;
        double gc2_c[E], gc3_c[E], fs_c = 0.;   // (cubic configs, taken early: cub_early)
#pragma unroll
        for (int e = 0; e < E; ++e) xea[e] = gc2_c[e] = gc3_c[e] = 0.;
        const bool evaluating = unit == U_EVAL;
        if (evaluating) {
            if (mode != M_OOB) {
                eps_t = (mode == M_LEAF) ? eps * (double)dir : 0.;
                const double dt = 0.5 * eps_t;
#pragma unroll
                for (int e = 0; e < E; ++e) p[e] = p[e] + dt * g[e];  // integration.py:80
                double vh[E];
                bool hit = false;
                if constexpr (VEL_AHEAD) {
                    if (ahead_ok) {
                        bool same = true;
#pragma unroll
                        for (int e = 0; e < E; ++e) same = same && (__double_as_longlong(p[e]) == __double_as_longlong(p_ahead[e]));
                        hit = __builtin_amdgcn_ballot_w64(!same) == 0ull;
                    }
                    ahead_ok = false;
                }
                if (hit) {
#pragma unroll
                    for (int e = 0; e < E; ++e) vh[e] = v_ahead[e];
                } else {
                    velocity(p, vh);                                   // :82
                }
#pragma unroll
                for (int e = 0; e < E; ++e) q[e] = q[e] + eps_t * vh[e];  // :85
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dim = lane * E + e;
                xo[e] = q[e];
                jac[e] = 1.;
                gj[e] = 0.;
                if (PLD && f_tr) {
                    // (the pipeline instantiation: one INLINED exponential and logarithm for every kind of bound -- the
                    // out-of-line libm calls of this file cost a round of register spills each at this kernel's pressure;
                    // the same expressions as bf_to_original_g of the lane-per-chain kernels)
                    const int kind = (int)pdl(PD_KIND, e);
                    const double rg = pdl(PD_RG, e);
                    const double ex = __ocml_exp_f64(kind == 1 ? -q[e] : q[e]);
                    const double t = 1. / (1. + ex);
                    double tmp = q[e], jt = 1., gq_ = 0.;
                    if (kind == 1) { tmp = t; jt = t * (1. - t); gq_ = (ex - 1.) * t; }
                    if (kind == 2) { tmp = ex; jt = ex; gq_ = 1.; }
                    if (kind == 3) { tmp = 1. - ex; jt = -ex; gq_ = 1.; }
                    xo[e] = pdl(PD_LO, e) + tmp * rg;
                    jac[e] = jt * rg;
                    gj[e] = gq_;
                    logdet += __ocml_log_f64(fabs(jac[e]));
                } else if (f_tr) {
                    double J, J2;
                    bf_to_original(q[e], (int)pdl(PD_KIND, e), pdl(PD_LO, e), pdl(PD_RG, e), xo[e], J, J2);
                    logdet += log(fabs(J));
                    jac[e] = J;
                    gj[e] = J2 / J;
                }
                xs[e] = f_su ? (xo[e] - pdl(PD_SU_LO, e)) / pdl(PD_SU_DIFF, e) : xo[e];
                double x_eval = xs[e];
                if (mode == M_OOB)  // modules/poly.py:482
                    x_eval = (m.alpha * xs[e] + (cs_get(CS_BETA) - m.alpha) * c_mu[e]) / cs_get(CS_BETA);
                xea[e] = x_eval;
                if (dim < DP) {
                    const int xi = (dim >> 2) * XS + w + 16 * (dim & 3);  // B[k = dim&3][n = chain] of k-step dim>>2
                    XB[0 * NS * XS + xi] = x_eval;
                    if (f_bound) XB[1 * NS * XS + xi] = xs[e] - c_mu[e];
                    if (f_decay) XB[2 * NS * XS + xi] = xo[e] - pdl(PD_DMU, e);
                    if constexpr (TAIL) {
                        XP[(0 * 16 + w) * DP + dim] = x_eval;
                        XP[(1 * 16 + w) * DP + dim] = xs[e] - c_mu[e];
                    }
                }
            }
        }
        // Bound proof at d = 128 (two jobs per wave: the H (x - mu) job is half of the tile phase): as in bfhip_group.h, the
        // test of modules/poly.py:467-469 is decided without those tiles while lam_max(H) |x - mu|^2 < alpha^2 holds for
        // every evaluating chain of the group; the outcome is the one the full computation has.
        constexpr bool PROOF = W == 8 && !FULLM;
        if constexpr (PROOF) {
            if (evaluating && f_bound) {
                double r2 = 0.;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double xm = xs[e] - c_mu[e];
                    r2 += (lane * E + e < d) ? pdl(PD_HD, e) * (xm * xm) : 0.;   // (the proof's weighted norm: bf_bound_lam_max_weighted)
                }
                r2 = wave_sum(r2);
                const bool inside = mode != M_OOB && g_sliced_proof_on(a) && m.lam_max * r2 < m.alpha * m.alpha * (1. - 1e-9);
                if (lane == 0 && !inside) alive[4 + (trip & 1)] = 1;
            }
        }
        TRACE(1);
        if (lane == 0) {
            if (unit != U_DONE) alive[trip & 1] = 1;
            if (TAIL && evaluating) atomicOr((unsigned *)&alive[2 + (trip & 1)], 1u << w);
            if (PLD && evaluating) alive[2 + (trip & 1)] = 1;   // (some chain of the workgroup needs the contractions in this trip)
        }
        stamp(0);
        __syncthreads();  // B1
        TRACE(2);
        stamp(1);
        if (rfl(alive[trip & 1]) == 0) break;  // every chain of the group is done (uniform)
        const unsigned ev_mask = TAIL ? (unsigned)rfl(alive[2 + (trip & 1)]) : 0u;
        // pipeline density: a trip in which NO chain of the workgroup evaluates (chains in step spend every second trip in a merge,
        // a doubling's end or the iteration's end) skips the matvec jobs and phase P with its three barriers -- uniform over the
        // workgroup, the flag is complete at B1
        const bool pld_eval = !PLD || rfl(alive[2 + (trip & 1)]) != 0;
        const bool skip_h = PROOF && f_bound && f_quad && !f_decay && rfl(alive[4 + (trip & 1)]) == 0;  // (uniform over the workgroup)
        if (tid == 0) { alive[(trip + 1) & 1] = 0; alive[2 + ((trip + 1) & 1)] = 0; alive[4 + ((trip + 1) & 1)] = 0; }
        TRACE(3);

        // ================= phase B: gradient tiles on MFMA =================
        // job = (enabled matrix, row tile t, K part): NS / KS k-steps of one 16-row tile for the 16 chains.
        // KS is chosen on the host so that each of the 16 waves gets a job whenever n_mat * W <= 8; the KS
        // partial results land in separate GB slots and are added by the chain's own wave in phase C.
        // All operands of a job (at most 8 k-steps at a time) are fetched from LDS before the first MFMA so
        // that the LDS latency is paid once per job, not once per k-step.
        auto run_jobs = [&](auto ks_tag, bool only_s = false) {
            constexpr int KS = decltype(ks_tag)::value;
            constexpr int KPJ = NS / KS;                        // k-steps per job
            constexpr int CH = KPJ < 8 ? KPJ : 8;               // k-steps fetched together
            const int mc = lane & 15, mg = lane >> 4;
            const bool quad = XT && cpg <= 4 && !g_no_quad_tiles(a);   // (fixed for the launch)
            // (skip_h: S and H are the only matrices and the H jobs, the second half of the list, are left out)
            const int n_job = (only_s || (skip_h && n_mat == 2 && mat0 == 0 && mat1 == 1)) ? W * KS : n_mat * (W * KS);
            if (cub_early) {
                // the chains' waves are busy with the cubic configs; the chain-less waves share the jobs, TWO at a time: two
                // independent chains of MFMAs whose operand loads (A from L2 at d = 128) are in flight together
                const int st = NWV - cpg;
                constexpr int C2 = KPJ < BF_JOB_CHUNK ? KPJ : BF_JOB_CHUNK;   // (k-steps whose operands are fetched together)
                for (int job = w - cpg; job >= 0 && job < n_job; job += 2 * st) {
                    const bool two = job + st < n_job;
                    const int jb2 = two ? job + st : job;
                    const int sm1 = job / (W * KS), r1 = job % (W * KS), t1 = r1 / KS, kp1 = r1 % KS;
                    const int sm2 = jb2 / (W * KS), r2 = jb2 % (W * KS), t2 = r2 / KS, kp2 = r2 % KS;
                    const int b1 = sm1 == 0 ? mat0 : (sm1 == 1 ? mat1 : 2), b2 = sm2 == 0 ? mat0 : (sm2 == 1 ? mat1 : 2);
                    const double *Af1 = (b1 == 0 ? Sf : (b1 == 1 ? Hf : Hdf)) + (t1 * NS + kp1 * KPJ) * 64 + lane;
                    const double *Af2 = (b2 == 0 ? Sf : (b2 == 1 ? Hf : Hdf)) + (t2 * NS + kp2 * KPJ) * 64 + lane;
                    const double *Xf1 = XB + (b1 * NS + kp1 * KPJ) * XS + lane, *Xf2 = XB + (b2 * NS + kp2 * KPJ) * XS + lane;
                    if (quad) {
                        // at most four chains in the workgroup: v_mfma_f64_4x4x4_4b -- four 4-row blocks of the tile against the
                        // SAME four columns (A lane 16 k + m as for the 16 x 16 x 4 tile, B lane 16 k + 4 b + n reads column n,
                        // D lane 16 i + 4 b + n is row 4 b + i of chain n) -- a quarter of the 16-column tile's time in the pipe
                        const int qo = (lane & ~15) + (lane & 3);
                        const double *Xq1 = Xf1 - lane + qo, *Xq2 = Xf2 - lane + qo;
                        double q1 = 0., q2 = 0.;
#pragma unroll
                        for (int c0 = 0; c0 < KPJ; c0 += C2) {
                            double a1[C2], x1[C2], a2[C2], x2[C2];
#pragma unroll
                            for (int q3 = 0; q3 < C2; ++q3) {
                                a1[q3] = Af1[(c0 + q3) * 64]; x1[q3] = Xq1[(c0 + q3) * XS];
                                a2[q3] = Af2[(c0 + q3) * 64]; x2[q3] = Xq2[(c0 + q3) * XS];
                            }
#pragma unroll
                            for (int q3 = 0; q3 < C2; ++q3) {
                                q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[q3], x1[q3], q1, 0, 0, 0);
                                q2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[q3], x2[q3], q2, 0, 0, 0);
                            }
                        }
                        const int qr = 4 * ((lane >> 2) & 3) + (lane >> 4), qn = lane & 3;
                        GB[((sm1 * KS + kp1) * 16 + qn) * GS + 16 * t1 + qr] = q1;
                        if (two) GB[((sm2 * KS + kp2) * 16 + qn) * GS + 16 * t2 + qr] = q2;
                        continue;
                    }
                    d4_t acc1 = {0., 0., 0., 0.}, acc2 = {0., 0., 0., 0.};
#pragma unroll
                    for (int c0 = 0; c0 < KPJ; c0 += C2) {
                        double a1[C2], x1[C2], a2[C2], x2[C2];
#pragma unroll
                        for (int q2 = 0; q2 < C2; ++q2) {
                            a1[q2] = Af1[(c0 + q2) * 64]; x1[q2] = Xf1[(c0 + q2) * XS];
                            a2[q2] = Af2[(c0 + q2) * 64]; x2[q2] = Xf2[(c0 + q2) * XS];
                        }
#pragma unroll
                        for (int q2 = 0; q2 < C2; ++q2) {
                            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q2], x1[q2], acc1, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[q2], x2[q2], acc2, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        GB[((sm1 * KS + kp1) * 16 + mc) * GS + 16 * t1 + 4 * r4 + mg] = acc1[r4];
                        if (two) GB[((sm2 * KS + kp2) * 16 + mc) * GS + 16 * t2 + 4 * r4 + mg] = acc2[r4];
                    }
                }
            } else
            for (int job = w; job < n_job; job += NWV) {
                if constexpr (AREG8 && KS == 1) {
                    // this wave's jobs, row tiles w and w + 4 of S and of H, as ONE pass over the k-steps with the A operands from
                    // registers: four independent accumulation chains (each the same instructions in the same order as the
                    // eight-wave form's job, so the same numbers); the next chunk's B operands are fetched under this chunk's MFMAs
                    if (mat0 == 0 && mat1 == 1) {
                        if (job != w) continue;   // (the wave's other jobs went with its first)
                        const bool with_h = n_job > W;
                        const double *Xs = XB + (lane & ~15) + (lane & 3), *Xh = Xs + NS * XS;
                        double q1 = 0., q2 = 0., q3 = 0., q4 = 0.;
                        double xs_[8], xh_[8];
#pragma unroll
                        for (int s = 0; s < 8; ++s) { xs_[s] = Xs[s * XS]; xh_[s] = with_h ? Xh[s * XS] : 0.; }
#pragma unroll
                        for (int c0 = 0; c0 < NS; c0 += 8) {
                            double ns_[8], nh_[8];
                            if (c0 + 8 < NS) {
#pragma unroll
                                for (int s = 0; s < 8; ++s) { ns_[s] = Xs[(c0 + 8 + s) * XS]; nh_[s] = with_h ? Xh[(c0 + 8 + s) * XS] : 0.; }
                            }
                            if (with_h) {
#pragma unroll
                                for (int s = 0; s < 8; ++s) {
                                    BF_MFMA_Q_ACC(q1, afr8[0][c0 + s], xs_[s]);
                                    BF_MFMA_Q_ACC(q2, afr8[1][c0 + s], xs_[s]);
                                    BF_MFMA_Q_ACC(q3, afr8[2][c0 + s], xh_[s]);
                                    BF_MFMA_Q_ACC(q4, afr8[3][c0 + s], xh_[s]);
                                }
                            } else {
#pragma unroll
                                for (int s = 0; s < 8; ++s) {
                                    BF_MFMA_Q_ACC(q1, afr8[0][c0 + s], xs_[s]);
                                    BF_MFMA_Q_ACC(q2, afr8[1][c0 + s], xs_[s]);
                                }
                            }
                            if (c0 + 8 < NS) {
#pragma unroll
                                for (int s = 0; s < 8; ++s) { xs_[s] = ns_[s]; xh_[s] = nh_[s]; }
                            }
                        }
                        BF_MFMA_Q_DONE(q1, q2, q3, q4);
                        double *gq = GB + (lane & 3) * GS + 4 * ((lane >> 2) & 3) + (lane >> 4);
                        gq[16 * w] = q1;
                        gq[16 * (w + 4)] = q2;
                        if (with_h) {
                            gq[16 * GS + 16 * w] = q3;
                            gq[16 * GS + 16 * (w + 4)] = q4;
                        }
                        TRACE(4);
                        TRACE(5);
                        continue;
                    }
                }
                const int slot_m = job / (W * KS), rem = job % (W * KS);
                const int t = rem / KS, kp = rem % KS;
                const int b = slot_m == 0 ? mat0 : (slot_m == 1 ? mat1 : 2);  // 0 S, 1 H, 2 H_decay
                const double *Af = (b == 0 ? Sf : (b == 1 ? Hf : Hdf)) + (t * NS + kp * KPJ) * 64 + lane;
                const double *Xf = XB + (b * NS + kp * KPJ) * XS + lane;
                if (quad) {   // (see above)
                    const double *Xq = Xf - lane + (lane & ~15) + (lane & 3);
                    double q1 = 0.;
#pragma unroll
                    for (int c0 = 0; c0 < KPJ; c0 += CH) {
                        double av[CH], xv[CH];
#pragma unroll
                        for (int s = 0; s < CH; ++s) { av[s] = Af[(c0 + s) * 64]; xv[s] = Xq[(c0 + s) * XS]; }
#pragma unroll
                        for (int s = 0; s < CH; ++s) q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av[s], xv[s], q1, 0, 0, 0);
                    }
                    GB[((slot_m * KS + kp) * 16 + (lane & 3)) * GS + 16 * t + 4 * ((lane >> 2) & 3) + (lane >> 4)] = q1;
                    continue;
                }
                d4_t acc = {0., 0., 0., 0.};
#pragma unroll
                for (int c0 = 0; c0 < KPJ; c0 += CH) {
                    double av[CH], xv[CH];
#pragma unroll
                    for (int s = 0; s < CH; ++s) { av[s] = Af[(c0 + s) * 64]; xv[s] = Xf[(c0 + s) * XS]; }
#pragma unroll
                    for (int s = 0; s < CH; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], xv[s], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) GB[((slot_m * KS + kp) * 16 + mc) * GS + 16 * t + 4 * r4 + mg] = acc[r4];
            }
        };
        // Tail of a launch: while at most tail_max chains of the group still evaluate, the 16-column MFMA tiles
        // are mostly padding and their fixed cost (32 MFMAs per SIMD) is the longest piece of the trip.  The same
        // numbers come from plain FMAs: v_mfma_f64_16x16x4_f64 accumulates each entry as ONE sequential fma chain
        // over k (checked bit for bit, tools/probe/mfma_arith_probe.hip), so lane = output row running that chain
        // over the K part reproduces the MFMA result exactly; wave w takes (chain w / WPC, matrix, K part).
        constexpr int WPC = 2 * KS_P, VMAX = 16 / WPC, KP = 4 * KPJ_P;
        const int n_ev = __builtin_popcount(ev_mask);
        if (TAIL && n_ev <= (a.tail_max < VMAX ? a.tail_max : VMAX)) {
            const int ci = w / WPC, b = (w / KS_P) & 1, kp = w % KS_P;
            if (ci < n_ev && lane < DP) {
                unsigned mm = ev_mask;
                for (int i = 0; i < ci; ++i) mm &= mm - 1;
                const int c = __builtin_ctz(mm);
                const double *Mr = RM + (b * DP + lane) * RS + kp * KP;
                const double *xc = XP + (b * 16 + c) * DP + kp * KP;
                double acc = 0.;
#pragma unroll
                for (int k = 0; k < KP; k += 2) {
                    const d2_t mv = *(const d2_t *)(Mr + k), xv2 = *(const d2_t *)(xc + k);
                    acc = __builtin_fma(mv[0], xv2[0], acc);
                    acc = __builtin_fma(mv[1], xv2[1], acc);
                }
                GB[((b * KS_P + kp) * 16 + c) * GS + lane] = acc;
            }
            TRACE(4);
            TRACE(5);
        } else if constexpr (AREG) {
            if (w < NJOB_P) {
                const int mc = lane & 15, mg = lane >> 4;
                const int slot_m = w / (W * KS_P), rem = w % (W * KS_P), t = rem / KS_P, kp = rem % KS_P;
                const double *Xf = XB + (slot_m * NS + kp * KPJ_P) * XS + lane;  // PLAIN: matrix id == slot
                d4_t acc = {0., 0., 0., 0.};
                double xv[KPJ_P];
#pragma unroll
                for (int s = 0; s < KPJ_P; ++s) xv[s] = Xf[s * XS];
#pragma unroll
                for (int s = 0; s < KPJ_P; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[s], xv[s], acc, 0, 0, 0);
                TRACE(4);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) GB[((slot_m * KS_P + kp) * 16 + mc) * GS + 16 * t + 4 * r4 + mg] = acc[r4];
            }
            TRACE(5);
        } else if (!pld_eval) {
        } else if (ks_rt == 2) run_jobs(std::integral_constant<int, (W >= 2 ? 2 : 1)>());
        else if (ks_rt == 4) run_jobs(std::integral_constant<int, (W >= 4 ? 4 : 1)>());
        else run_jobs(std::integral_constant<int, 1>());
        if (cub_early && evaluating) { TRACE(4); fs_c = cubic_lds(xea, gc2_c, gc3_c); TRACE(5); }   // (this wave had no job: see run_jobs)
        const int unit_in = unit;
        stamp(2);
        __syncthreads();  // B2
        TRACE(6);
        stamp(1);

        // ================= phase C: finish the evaluation =================
        double gn[E], hv[E], dgr[E];
        double logp_new = 0., E_new = 0.;
        bool have_eval = false, kin_ready = false;
        double kin_fast = 0.;
        if constexpr (PLD) {
            // ================= phase P: the pipeline density (bfhip_pld.h) =================
            // Every wave of the workgroup takes part in the two contractions; a chain that evaluates owns column w of them.
            // The bound is decided FIRST (its H (x - mu) tiles are this trip's phase B), so a point outside the ellipsoid is
            // evaluated once, at its projection (modules/poly.py:480-503), never in a second trip.
            const PldDev &pl = m.pld;
            double xmv[E], beta_o = 0., r_bd2 = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e) xmv[e] = 0.;
            if (evaluating) {
                double r_b2 = 0.;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int dim = lane * E + e;
                    hv[e] = (f_bound && lane_ok) ? gb_read(slot_H, dim) : 0.;
                    dgr[e] = (f_decay && lane_ok) ? gb_read(slot_D, dim) : 0.;
                    xmv[e] = xs[e] - c_mu[e];
                    r_b2 += xmv[e] * hv[e];
                    if (f_decay) r_bd2 += (xo[e] - pdl(PD_DMU, e)) * dgr[e];
                }
                { double r2[2] = {r_b2, r_bd2}; wave_sum_n<2>(r2); r_b2 = r2[0]; r_bd2 = r2[1]; }
                if (f_tr) logdet = wave_sum(logdet);
                if (f_bound && !(r_b2 < m.alpha * m.alpha * (1. - 1e-12))) {   // modules/poly.py:467-469
                    const double b = usqrt(r_b2);
                    if (b > m.alpha) beta_o = b;
                }
                double x_ev[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double xv = beta_o > 0. ? (m.alpha * xs[e] + (beta_o - m.alpha) * c_mu[e]) / beta_o : xs[e];   // :482
                    x_ev[e] = lane * E + e < d ? xv : 0.;
                }
                pld_point_e<E>(pl, PL, DP, w, lane, x_ev, beta_o);
            }
            if (pld_eval) {   // (uniform over the workgroup)
            TRACEP(7);
            __syncthreads();  // P1: monomials of every evaluating chain
            TRACEP(8);
            if constexpr (NWV == 8) pld_gemm1_q8(pl, PL, m.alpha, w, NWV, lane);   // (at most eight chains: 4 x 4 x 4 tiles)
            else pld_gemm1_w16(pl, PL, m.alpha, w, NWV, lane);
            TRACEP(11);
            __syncthreads();  // P2: residuals
            TRACEP(12);
            if constexpr (NWV == 8) pld_gemm2_q8(pl, PL, w, NWV, lane);
            else pld_gemm2_w16(pl, PL, w, NWV, lane);
            TRACEP(13);
            __syncthreads();  // P3: W = C'^T r
            TRACEP(14);
            }
            if (evaluating) {
                double s2[2];
                pld_sums(pl, PL, w, lane, NWV, s2[0], s2[1]);
                wave_sum_n<2>(s2);
                double gj0[E], dj = 0.;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    gj0[e] = lane * E + e < DP ? pld_grad(pl, PL, DP, w, lane * E + e) : 0.;   // (J_0^T r)_dim
                    dj += gj0[e] * xmv[e];
                }
                if (beta_o > 0.) {   // (compressed outputs: the tails of Q^T f_mu' and Q^T y' as scalars, bfhip_pipeline_upload)
                    const double b = (beta_o - m.alpha) / m.alpha;
                    s2[0] += b * (b * pl.k_ff + 2. * pl.k_fy);
                    s2[1] += b * pl.k_ff + pl.k_fy;
                }
                TRACEP(15);
                if (beta_o > 0.) {   // modules/poly.py:494-496, contracted with r
                    const double r_dotj = wave_sum(dj);
#pragma unroll
                    for (int e = 0; e < E; ++e) gj0[e] += (s2[1] / m.alpha - r_dotj / beta_o) * (hv[e] / beta_o);
                }
                double f = pl.logp0 - 0.5 * s2[0];
                double pr = 0.;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int dim = lane * E + e;
                    gn[e] = -gj0[e];                                   // density.py:552-560: dot(J_like, J_surrogate)
                    if (f_su) gn[e] = gn[e] / pdl(PD_SU_DIFF, e);      // module.py:226
                    gn[e] = gn[e] * jac[e];                            // density.py:558
                    if (pl.has_prior) {   // the last module: like + log prior of the original-space inputs
                        const double dx = dim < d ? xo[e] - pl.prior_mu[dim] : 0., pp = dim < d ? pl.prior_prec[dim] : 0.;
                        pr += pp * dx * dx;
                        gn[e] += -(pp * dx) * jac[e];
                    }
                }
                if (pl.has_prior) f += pl.prior_c0 - 0.5 * wave_sum(pr);
                if (f_decay) {
                    f -= m.decay_gamma * bf_clip0(r_bd2 - m.decay_alpha2);
                    if (r_bd2 > m.decay_alpha2) {
#pragma unroll
                        for (int e = 0; e < E; ++e) gn[e] -= 2. * m.decay_gamma * dgr[e];
                    }
                }
                if (f_tr) {
                    f += logdet;
#pragma unroll
                    for (int e = 0; e < E; ++e) gn[e] += gj[e];
                }
                logp_new = f;
                have_eval = true;
            }
        } else
        if (evaluating) {
            double r_quad = 0., r_lin = 0., r_b2 = 0., r_dotj = 0., r_bd2 = 0., r_kin = 0.;
            const bool fast_kin = !FULLM && !f_decay && !f_link && mode != M_OOB;
            double xev[E], sxv[E], r_cub = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dim = lane * E + e;
                const double sx = (f_quad && lane_ok) ? gb_read(slot_S, dim) : 0.;
                sxv[e] = sx;
                hv[e] = (f_bound && lane_ok && !skip_h) ? gb_read(slot_H, dim) : 0.;
                dgr[e] = (f_decay && lane_ok) ? gb_read(slot_D, dim) : 0.;
                xev[e] = xs[e];
                if (mode == M_OOB) xev[e] = (m.alpha * xs[e] + (cs_get(CS_BETA) - m.alpha) * c_mu[e]) / cs_get(CS_BETA);
                if constexpr (SPEC) {
                    // value of the linear + quadratic surrogate summed per lane: one reduction instead of two (the
                    // same expression, spelled with an explicit fma, in bf_nuts_pipe_kernel)
                    r_lin += __builtin_fma(0.5 * xev[e], sx, c_lin[e] * xev[e]);
                } else {
                    r_quad += xev[e] * sx;
                    r_lin += c_lin[e] * xev[e];
                }
                gn[e] = sx + c_lin[e];
            }
            if (f_cubic) {
                // Cubic configs (modules/_poly.pyx:49-137), the whole wave on one chain's terms.  x_k of this chain is lane
                // k / E, element k % E.  The work is laid out over (j, k): lane (jl = lane & 15, kq = lane >> 4) accumulates, for
                // output index j = jb + jl, the terms with k = kq, kq + 4, ... -- the coefficient tables are stored with j
                // contiguous, so a row of 16 lanes reads 128 contiguous bytes -- the four kq parts are added by two row
                // swaps, and the lane that owns dimension mask[j] fetches its result.  (bf_cubic_grad, the per-dimension
                // form, left 8 of 64 lanes with n^2 serial iterations each at config 5's 16 masked inputs.)
                const int jl = lane & 15, kq = lane >> 4;
                auto xu = [&](int dim) { return readlane_f64((E > 1 && (dim % E)) ? xev[E - 1] : xev[0], dim / E); };  // wave-uniform
                auto xl = [&](int dim) {  // per-lane dimension
                    const double a0 = __shfl(xev[0], dim / E, 64);
                    if constexpr (E > 1) { const double a1 = __shfl(xev[E - 1], dim / E, 64); return (dim % E) ? a1 : a0; }
                    return a0;
                };
                auto fetch = [&](const int *pos, int jb, double val, double (&dst)[E], bool add_value, double fval) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int dim = lane * E + e;
                        const int pj = dim < m.DP ? pos[dim] : -1;
                        const bool mine = pj >= jb && pj < jb + 16;
                        const double gv = __shfl(val, mine ? pj - jb : 0, 64);
                        if (mine) dst[e] += gv;
                    }
                    (void)add_value; (void)fval;
                };
                double fsum = 0.;
                if (__builtin_expect(cub_l, 1)) {
                    // the same sums in the same order with the tables in LDS, the masked inputs gathered once (lane k
                    // holds x[mask[k]]) and the masks and positions of this lane in registers: nothing of the loops below
                    // goes to global memory (the 64 dependent global loads of the cubic-3 contraction were 2/3 of config 5's trip)
                    // (cubic_lds above: the sums and their order are those of the separate loops below)
                    if (!cub_early) fs_c = cubic_lds(xev, gc2_c, gc3_c);
#pragma unroll
                    for (int e = 0; e < E; ++e) { gn[e] += gc2_c[e]; gn[e] += gc3_c[e]; }
                    fsum = fs_c;
                } else {
                for (int jb = 0; jb < m.n2; jb += 16) {   // cubic-2: f = sum_j x_j^2 v1_j, v1 = A x; df/dx_j = 2 x_j v1_j + (A^T x^2)_j
                    const int j = jb + jl;
                    const bool on = j < m.n2;
                    double v1 = 0., v2 = 0.;
                    for (int kk = 0; kk < m.n2; kk += 4) {   // (a wave-uniform loop; k differs between the rows of the wave)
                        const int k = kk + kq;
                        const bool ok = on && k < m.n2;
                        const double xk = xl(ok ? m.mask2[k] : 0);
                        v1 += (ok ? m.A2t[k * m.n2 + j] : 0.) * xk;
                        v2 += (ok ? m.A2[k * m.n2 + j] : 0.) * (xk * xk);
                    }
                    v1 = swap32_add_f64(swap16_add_f64(v1));
                    v2 = swap32_add_f64(swap16_add_f64(v2));
                    const double xj = xl(on ? m.mask2[j] : 0);
                    const double gj2 = 2. * xj * v1 + v2;
                    if (on && kq == 0) fsum += xj * xj * v1;
                    fetch(m.pos2, jb, gj2, gn, false, 0.);
                }
                for (int jb = 0; jb < m.n3; jb += 16) {   // cubic-3: df/dx_j = 1/2 sum_{k,l} T[j,k,l] x_k x_l, f = x . grad / 3
                    const int j = jb + jl;
                    const bool on = j < m.n3;
                    double sacc = 0.;
                    for (int kk = 0; kk < m.n3; kk += 4) {
                        const int k = kk + kq;
                        const bool ok = on && k < m.n3;
                        double t = 0.;
                        const double *Tk = m.T3t + (size_t)(ok ? k : 0) * m.n3 * m.n3 + (ok ? j : 0);
                        for (int l = 0; l < m.n3; ++l) t += (ok ? Tk[(size_t)l * m.n3] : 0.) * xu(rfl(m.mask3[l]));
                        sacc += t * xl(ok ? m.mask3[k] : 0);
                    }
                    sacc = swap32_add_f64(swap16_add_f64(sacc));
                    const double xj = xl(on ? m.mask3[j] : 0);
                    if (on && kq == 0) fsum += xj * (0.5 * sacc) * (1. / 3.);
                    fetch(m.pos3, jb, 0.5 * sacc, gn, false, 0.);
                }
                }
                r_cub = wave_sum(fsum);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double xm = xs[e] - c_mu[e];
                r_b2 += xm * hv[e];
                r_dotj += gn[e] * xm;  // dot(jj_0, x - mu), poly.py:496 (used in the OOB pass only)
                if (f_decay) r_bd2 += (xo[e] - pdl(PD_DMU, e)) * dgr[e];
                if (fast_kin) {  // in-bound gradient is already final: the kinetic energy rides along
                    double ge = gn[e];
                    if (f_su) ge = ge / pdl(PD_SU_DIFF, e);
                    ge = ge * jac[e];
                    if (f_tr) ge += gj[e];
                    const double pe = p[e] + (0.5 * eps_t) * ge;
                    r_kin += pe * (var[e] * pe);
                }
            }
            TRACE(7);
            if constexpr (SPEC) {  // the reductions every evaluation needs, advanced together
                double r3[3] = {r_kin, r_lin, r_b2};
                wave_sum_n<3>(r3);
                r_kin = r3[0]; r_lin = r3[1]; r_b2 = r3[2];
            } else {
                double r4[4] = {r_kin, r_quad, r_lin, r_b2};
                wave_sum_n<4>(r4);
                r_kin = r4[0]; r_quad = r4[1]; r_lin = r4[2]; r_b2 = r4[3];
            }
            TRACE(8);
            if (f_bound && mode == M_OOB) r_dotj = wave_sum(r_dotj);
            if (f_decay) r_bd2 = wave_sum(r_bd2);
            if (f_tr) logdet = wave_sum(logdet);

            double f = SPEC ? (m.c0 + r_lin) + r_cub : ((m.c0 + r_lin) + 0.5 * r_quad) + r_cub;
            // beta = sqrt(r_b2) is only needed outside the ellipsoid; the test beta > alpha (poly.py:467-469) is
            // decided on the squares whenever r_b2 is not within rounding distance of alpha^2, so the common
            // in-bound evaluation has no sqrt on its critical path and the decision is still the reference's
            double beta = 0.;
            bool oob_now = false;
            if (f_bound) {
                const double a2 = m.alpha * m.alpha;
                if (mode != M_OOB && !(r_b2 < a2 * (1. - 1e-12))) beta = usqrt(r_b2);
                if (mode == M_OOB) {  // second pass: f, gn currently hold f_0 and jj_0 (poly.py:484-496)
                    const double f0 = f, beta_saved = cs_get(CS_BETA);
                    f = (beta_saved * f0 - (beta_saved - m.alpha) * m.f_mu) / m.alpha;
                    const double coef = (f0 - m.f_mu) / m.alpha - r_dotj / beta_saved;
#pragma unroll
                    for (int e = 0; e < E; ++e) gn[e] = gn[e] + coef * (hv[e] / beta_saved);
                } else if (beta > m.alpha) {
                    oob_now = true;
                }
            }
            bool oob_lin = false;
            if (oob_now && !f_cubic) {
                // linear + quadratic surrogate: S x_0 follows from S x, no second pass (bfhip_oob.h; the same expressions in
                // bf_nuts_pipe_kernel)
                double r2[2] = {0., 0.};
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double xm = xs[e] - c_mu[e], smu = pdl(PD_SMU, e);
                    r2[0] += xm * (smu + c_lin[e]);
                    r2[1] += xm * (sxv[e] - smu);
                }
                wave_sum_n<2>(r2);
                const BfOob o = bf_oob_scalars(m.alpha, m.inv_alpha, m.f_mu, m.f_poly_mu, beta, r2[0], r2[1]);
                f = o.f;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double smu = pdl(PD_SMU, e);
                    gn[e] = bf_oob_grad(o, smu + c_lin[e], sxv[e] - smu, hv[e]);
                }
                oob_now = false;
                oob_lin = true;
            }
            kin_ready = fast_kin && !oob_now && !oob_lin;
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
                    if (f_su) gn[e] = gn[e] / pdl(PD_SU_DIFF, e);
                    gn[e] = gn[e] * jac[e];
                }
                if (f_link) {  // logp = phi(m), grad = phi'(m) grad m
                    const double r = f - m.link_y, dphi = -(m.link_prec * r);
                    f = m.link_logp0 - 0.5 * (r * (m.link_prec * r));
#pragma unroll
                    for (int e = 0; e < E; ++e) gn[e] = dphi * gn[e];
                }
                if (f_decay) {
                    f -= m.decay_gamma * bf_clip0(r_bd2 - m.decay_alpha2);
                    if (r_bd2 > m.decay_alpha2) {
#pragma unroll
                        for (int e = 0; e < E; ++e) gn[e] -= 2. * m.decay_gamma * dgr[e];
                    }
                }
                if (f_tr) {
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
                g[e] = gn[e];
            }
            // (not while the metric adapts: an iteration then ends with passes over three more matrices and the second column costs
            // more than the pass it saves -- 3.0 against 3.3 x 10^7, tools/full_metric_rate.py)
            if constexpr (VEL_AHEAD) {
                // (integration.py:92, and :82 of the step that follows if the tree goes on from here: p + eps/2 g with this step's eps)
#pragma unroll
                for (int e = 0; e < E; ++e) p_ahead[e] = p[e] + dt * g[e];
                bf_velocity_full2<E>(matp + BF_MAT_COV * msz, p, p_ahead, vcur, v_ahead, d, lane);
                ahead_ok = mode != M_INIT;   // (the evaluation that opens a launch is a step of length 0)
#pragma unroll
                for (int e = 0; e < E; ++e) kin += p[e] * vcur[e];
            } else if constexpr (FULLM) {
                velocity(p, vcur);               // integration.py:92
#pragma unroll
                for (int e = 0; e < E; ++e) kin += p[e] * vcur[e];
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) kin += p[e] * (var[e] * p[e]);   // metrics.py:88-91
            }
            kin = kin_ready ? kin_fast : wave_sum(kin);
            E_new = 0.5 * kin - logp_new;        // integration.py:92-93
        }

        stamp(3);
        TRACE(9);
        const int mode_in = mode;
        // every chain runs ONE unit here, in parallel: chains that evaluated finish their leaf / init, the
        // others do their pending merge level / doubling end / iteration-end piece
        if constexpr (CHAIN_UNITS) {
            // Expensive trips (the pipeline density: two contractions and three more barriers, tens of microseconds; d = 128: 12 us):
            // a bookkeeping unit is a few hundred cycles next to that, so the chain runs its units until it needs the next
            // gradient (or is done) instead of spending a trip on each -- 8 trips per 7-leaf iteration instead of 15, two thirds of
            // the trips of a 1023-leaf tree.  The same units in the same order: samples, statistics and random streams do not
            // change (the prefetches a unit issues for its successor are simply consumed at once).  Measured: DES-shaped pipeline
            // 7.5 -> 9.8 x 10^7 leapfrog steps/s, config 5 (d = 128, 1023-leaf trees) 7.4 -> 8.2, full-rank metric 7.0 -> 7.4 fixed
            // and 2.1 -> 2.5 adapting.  (At d <= 64 on the common surrogate a trip is 3 us and one unit per trip was measured
            // faster, see run_unit.)
            bool first = true;
            do {
                run_unit(first && unit_in == U_EVAL && have_eval, E_new, logp_new);
                first = false;
            } while (unit != U_EVAL && unit != U_DONE);
        } else
        if (unit_in == U_EVAL) run_unit(have_eval, E_new, logp_new);
        else run_unit(false, 0., 0.);
        TRACE(10);
        stamp(unit_in == U_EVAL ? (mode_in == M_INIT ? 4 : 5) : (unit_in == U_MERGE_RUN ? 6 : (unit_in == U_DBL_END ? 7 : (unit_in == U_DONE ? 9 : 8))));
    }

#ifdef BF_TRACE
    if (a.stamps && w == 0 && blockIdx.x == 0)
        for (int i = lane; i < BF_TRACE * 16; i += 64) a.stamps[i] = TRC[i];
#endif
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
            scp[BFHIP_SC_LOG_STEP] = cs_get(CS_LOG_STEP);
            scp[BFHIP_SC_LOG_BAR] = cs_get(CS_LOG_BAR);
            scp[BFHIP_SC_HBAR] = cs_get(CS_HBAR);
            scp[BFHIP_SC_COUNT] = cs_get(CS_COUNT);
            scp[BFHIP_SC_I_ITER] = (double)i_iter;
            scp[BFHIP_SC_ERROR] = (double)err;
            if (a.n_leapfrog && nlf) atomicAdd(a.n_leapfrog, nlf);
        }
    }
}

#ifdef BF_TRACE   // (the macro above names PLD, a template constant of bf_sampler_kernel)
#undef TRACE
#define TRACE(k) do { if (w == 0 && blockIdx.x == 0 && trip_no < BF_TRACE && lane == 0) TRC[trip_no * 16 + (k)] = clock64(); } while (0)
#endif
#include "bfhip_nuts_pipe.h"
#include "bfhip_lone.h"

// the common surrogate: linear + quadratic configs with the extrapolation bound and nothing else
static bool sampler_plain(const DevModel &m) {
    return m.has_quad && m.use_bound && !m.use_decay && !m.has_transform && !m.has_su && !m.has_cubic && !m.has_link && !(bf_tune().no_plain != 0);
}

// K-split of the matvec jobs: the largest power of two KS <= W with n_mat * W * KS <= 16
static int sampler_ksplit(const DevModel &m) {
    const int W = m.DP / 16;
    const int n_mat = (m.has_quad ? 1 : 0) + (m.use_bound ? 1 : 0) + (m.use_decay ? 1 : 0);
    int ks = 1;
    while (2 * ks <= W && n_mat * W * 2 * ks <= 16) ks *= 2;
    return ks;
}
static int sampler_gb_slots(const DevModel &m) {
    const int n_mat = (m.has_quad ? 1 : 0) + (m.use_bound ? 1 : 0) + (m.use_decay ? 1 : 0);
    return n_mat * sampler_ksplit(m) > 1 ? n_mat * sampler_ksplit(m) : 1;
}

static size_t sampler_lds_base(const DevModel &m, bool plain) {
    const int W = m.DP / 16, DP = m.DP, NS = 4 * W;
    size_t dbl = (size_t)3 * NS * 65 + (size_t)sampler_gb_slots(m) * 16 * (DP + 1) + (size_t)16 * BFHIP_MAX_TREEDEPTH * LS_N + 4 +
                 (size_t)16 * CS_N + (size_t)PD_N * DP;
    if (DP <= 64 && plain)  // A operands in registers; row-major S, H and plain x for the VALU matvec
        dbl += (size_t)2 * DP * (DP + 2) + (size_t)2 * 16 * DP;
    else if (DP <= 64)
        dbl += (size_t)DP * DP * ((m.has_quad ? 1 : 0) + (m.use_bound ? 1 : 0) + (m.use_decay ? 1 : 0));
    return dbl;
}
// cubic coefficient tables in LDS (bf_sampler_kernel: cub_l): when both masks fit a wave and the tables fit behind the rest
static size_t sampler_cubic_doubles(const DevModel &m) {
    return (size_t)2 * m.n2 * m.n2 + (size_t)((m.n3 + 15) >> 4) * m.n3 * m.n3 * 16 + 2;   // (bf_sampler_kernel: T3x and its alignment)
}
static bool sampler_cubic_lds(const DevModel &m, bool plain) {
    return m.has_cubic && m.n2 <= 64 && m.n3 <= 64 &&
           (sampler_lds_base(m, plain) + sampler_cubic_doubles(m)) * sizeof(double) <= (size_t)160 * 1024;
}
static size_t sampler_lds_bytes(const DevModel &m, bool plain, int nwv = 16) {
    if (m.pld.on)   // (the pipeline block sits where the cubic tables would: behind the matvec results, 16-byte aligned)
        return (((sampler_lds_base(m, false) + 1) & ~(size_t)1) +
                pld_lds_doubles(m.DP, m.pld.MP, m.pld.PP, m.pld.KS2, m.pld.n_ent, nwv == 8 ? PLD_XS8 : PLD_XS)) * sizeof(double);
    return (sampler_lds_base(m, plain) + (sampler_cubic_lds(m, plain) ? sampler_cubic_doubles(m) : 0)) * sizeof(double);
}
// what the sampler's own regions take for a pipeline density (bfhip_pipeline_upload sizes the K-split of GEMM2 with it)
size_t bf_sampler_lds_bytes_base(const DevModel &m) { return ((sampler_lds_base(m, false) + 1) & ~(size_t)1) * sizeof(double); }

// Chains per workgroup of the wave-per-chain kernels.  A launch lasts (trips of its longest chain) x (time of a trip), and a
// trip is its matvec jobs -- the same MFMAs whatever the number of columns in use -- plus the bookkeeping of the chains'
// waves, which share four SIMDs.  When the chains do not fill the chip at 16 per workgroup, fewer chains per workgroup
// on more CUs shorten the trip: the waves without a chain still take their share of the jobs.  (Results do not depend on
// it: a chain's arithmetic never involves its neighbours'.)  BFHIP_WAVE_CPG / bfhip_debug_set("wave_cpg") override (tests, tuning).
static int wave_layout_cpg(const bfhip_ctx *ctx, int n_chain, int nwv) {
    const int forced = bf_tune().wave_cpg;
    if (forced > 0) return forced < nwv ? forced : nwv;
    int cpg = nwv;
    while (cpg > 1 && (n_chain + cpg / 2 - 1) / (cpg / 2) <= ctx->n_cu) cpg /= 2;
    return cpg;
}

template <int W, bool NUTS, bool STAMPS, int FS, int FULLM = 0>
static int launch_sampler_t(bfhip_ctx *ctx, const SamplerArgs &args_in) {
    auto k = bf_sampler_kernel<W, NUTS, STAMPS, FS, FULLM>;
    constexpr int NWV = BF_SAMPLER_WAVES(W, FULLM, FS);
    size_t lds = sampler_lds_bytes(ctx->model, FS == 1, NWV);
    SamplerArgs args = args_in;
    args.pld_cl = 0;
    if (ctx->model.pld.on && NWV == 8 && !(bf_tune().pld_no_cl != 0)) {
        // the eight-wave forms read the A operands of both contractions from a row-major copy of C' in LDS when it fits behind the
        // rest (the DES shape: 52 KB), instead of streaming 2 x 51 KB of fragments from L2 in every trip -- the same numbers
        const size_t with_cl = lds + pld_cl_doubles(ctx->model.pld.MP, ctx->model.pld.PP) * sizeof(double) + 16;
        if (with_cl <= (size_t)160 * 1024) { lds = with_cl; args.pld_cl = 1; }
    }
    if (lds > 64 * 1024)
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    args.cpg = wave_layout_cpg(ctx, args.n_chain, NWV);
    args.cub_lds = sampler_cubic_lds(ctx->model, FS == 1) ? 1 : 0;
    args.cub_loops = bf_tune().cubic_loops;
    const int groups = (args.n_chain + args.cpg - 1) / args.cpg;
    snprintf(bf_tune().last_kernel, sizeof(bf_tune().last_kernel), "bf_sampler_kernel<%d, %s, %s, %d, %d>", W, NUTS ? "true" : "false",
             STAMPS ? "true" : "false", FS, FULLM);
    hipLaunchKernelGGL(k, dim3(groups), dim3(NWV * 64), lds, ctx->stream, ctx->model, args);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// test / tuning hook: NUTS on the common surrogate through bf_sampler_kernel instead of the pipelined kernel
// (also selected by the environment variable BFHIP_NUTS_KERNEL=sliced)

// The tail of a launch.  A launch of the wave-per-chain kernel lasts as long as its busiest chain, and at its end most workgroups
// hold one or two unfinished chains on 16-column tiles.  Sixteen-chain launches therefore run in two parts: in the first a
// workgroup with at most four unfinished chains lets each of them stop at the end of its iteration (tail_stop) -- once three
// quarters of the launch's chains are through and the chain has at least a quarter of the launch's iterations left (a stopped
// chain waits for the first part to end: worth it for a straggler in the tail, not for a chain that is merely last); a small kernel
// lists the chains that have iterations left; the second part runs those, one to four per workgroup on 4 x 4 x 4 tiles (a lone
// chain's leapfrog step: 2.97 against 3.90 us, tools/lone_funnel.py).  A chain's numbers depend neither on where it is cut (as
// between any two launches) nor on its workgroup.  bfhip_debug_set("tail_relaunch", 0) / BFHIP_TAIL_RELAUNCH=0: one part (tests compare).
// test hook: how many chains the last two-part launch listed for its second part (synchronises), -1 without one
extern "C" int bfhip_debug_tail_count(bfhip_ctx *ctx) {
    if (!ctx || !ctx->tail_buf) return -1;
    int n = -1;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
    if (hipMemcpy(&n, ctx->tail_buf, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
static int tail_stop_now() {
    const int hi = bf_tune().lone ? 16 : 4;   // (the few-chain instantiation of the second part takes four chains of every first-part workgroup at most)
    return bf_tune().tail_stop < 1 ? 1 : (bf_tune().tail_stop > hi ? hi : bf_tune().tail_stop);
}

__global__ void bf_tail_list_kernel(int n_chain, int iter_end, const double *sc, int *buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chain) return;
    const double *scp = sc + (size_t)i * BFHIP_SC_N;
    if ((int)scp[BFHIP_SC_I_ITER] < iter_end && (int)scp[BFHIP_SC_ERROR] == 0) buf[2 + atomicAdd(buf, 1)] = i;   // buf: count | done | list
}


// The latency kernel (bfhip_lone.h): one chain per workgroup of 2 + W waves.  It serves the second part of a two-part launch
// (the chains listed in tail_buf) and whole launches whose chains all fit the chip at once -- every workgroup must be resident,
// a chain that waited for a slot would double the launch.  bf_tune().lone: 1 automatic, 0 never, 2 wherever it is implemented
// (tests: any chain count, in as many rounds as it takes).


// the second instantiation's waves per SIMD: 4 (W + 1) waves a CU at d <= 32, 2 x 5 at d <= 64
template <int W> struct LoneOcc { static constexpr int MINW = W == 1 ? 2 : 3; };   // (W = 4: two workgroups of five waves a CU)

// form 0: the roomy instantiation (W job waves at d > 32); form 1: three job waves (d > 32: two four-wave workgroups a CU at 256
// registers); form 2: the tight instantiation (168 / 128 registers: the tail's stragglers when they outnumber the CUs' room)
template <int W, bool TR, int DEC, int FORM>
static const void *lone_kernel_ptr() {
    if constexpr (FORM == 0) { auto k = bf_lone_kernel<W, TR, DEC, 1, 0>; return (const void *)k; }
    else if constexpr (FORM == 1) { auto k = bf_lone_kernel<W, TR, DEC, 1, (W == 4 ? 1 : 0)>; return (const void *)k; }
    else { auto k = bf_lone_kernel<W, TR, DEC, LoneOcc<W>::MINW, 0>; return (const void *)k; }
}
template <int W, int DEC, int FORM> constexpr int lone_threads() { return LoneWaves<W, DEC, (FORM == 1 && W == 4) ? 1 : 0>::NW * 64; }

template <int W, bool TR, int DEC, int FORM>
static int lone_blocks_per_cu(bfhip_ctx *ctx) {
    // (asked once per instantiation and device, not at every launch: two runtime calls each)
    static int cached[64];
    static bool have[64];
    const int dev = (ctx->device >= 0 && ctx->device < 64) ? ctx->device : 0;
    if (have[dev]) return cached[dev];
    const size_t lds = LoneGeo<W, DEC>::n_doubles * sizeof(double);
    int nb = 0;
    const void *k = lone_kernel_ptr<W, TR, DEC, FORM>();
    if (lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, lone_threads<W, DEC, FORM>(), lds) != hipSuccess) return 0;
    cached[dev] = nb;
    have[dev] = true;
    return nb;
}

template <int W, bool TR, int DEC, int FORM>
static void lone_launch_form(bfhip_ctx *ctx, const SamplerArgs &args, int n_blocks) {
    const size_t lds = LoneGeo<W, DEC>::n_doubles * sizeof(double);
    if constexpr (FORM == 0) hipLaunchKernelGGL((bf_lone_kernel<W, TR, DEC, 1, 0>), dim3(n_blocks), dim3(lone_threads<W, DEC, 0>()), lds, ctx->stream, ctx->model, args);
    else if constexpr (FORM == 1) hipLaunchKernelGGL((bf_lone_kernel<W, TR, DEC, 1, (W == 4 ? 1 : 0)>), dim3(n_blocks), dim3(lone_threads<W, DEC, 1>()), lds, ctx->stream, ctx->model, args);
    else hipLaunchKernelGGL((bf_lone_kernel<W, TR, DEC, LoneOcc<W>::MINW, 0>), dim3(n_blocks), dim3(lone_threads<W, DEC, 2>()), lds, ctx->stream, ctx->model, args);
}

// returns 1 when the launch was taken, 0 when the caller should use the pipelined kernel, < 0 on error
template <int W, bool TR, int DEC>
static int launch_lone(bfhip_ctx *ctx, const SamplerArgs &args_in, int n_blocks, bool tail) {
    if (!bf_tune().lone || args_in.stamps) return 0;
    SamplerArgs args = args_in;
    args.n_cu = ctx->n_cu;
    args.tail_stop = 0;
    args.tail_done = NULL;
    if (!tail) { args.tail_list = NULL; args.tail_count = NULL; }
    args.stamps = bf_tune().stamps_lone;
    // every workgroup must be resident (a chain that waited for a slot would double the launch): the roomiest form that holds them
    const int need = (n_blocks + ctx->n_cu - 1) / ctx->n_cu;
    int form = 0;
    if (lone_blocks_per_cu<W, TR, DEC, 0>(ctx) < need) {
        form = 1;
        if (W != 4 || lone_blocks_per_cu<W, TR, DEC, 1>(ctx) < need) {
            form = 2;
            // (a whole launch takes the kernel only in a roomy form: two workgroups per CU at 168 registers ran a 64-d chain at half
            // the speed of one; the tight form is for the tail, whose chains are there anyway, and for tests)
            if (bf_tune().lone != 2 && !tail) return 0;
        }
    }
    if (bf_tune().lone_form >= 0 && bf_tune().lone_form <= 2) form = bf_tune().lone_form;   // (tests: the forms give the same numbers)
    if (form == 0) lone_launch_form<W, TR, DEC, 0>(ctx, args, n_blocks);
    else if (form == 1) lone_launch_form<W, TR, DEC, 1>(ctx, args, n_blocks);
    else lone_launch_form<W, TR, DEC, 2>(ctx, args, n_blocks);
    BF_HIP_CHECK(hipGetLastError());
    if (!tail) snprintf(bf_tune().last_kernel, sizeof(bf_tune().last_kernel), "bf_lone_kernel<%d, %s, %d, %d>", W, TR ? "true" : "false", DEC, form);
    return 1;
}

template <int W, bool TR = false, int DEC = 0>
static int launch_nuts_pipe(bfhip_ctx *ctx, const SamplerArgs &args_in) {
    SamplerArgs args = args_in;
    args.cpg = wave_layout_cpg(ctx, args.n_chain, 16);
    args.tail_stop = 0;
    args.tail_list = NULL;
    args.tail_count = NULL;
    args.tail_done = NULL;
    args.n_cu = ctx->n_cu;
    // (at most four / eight chains in a workgroup: 4 x 4 x 4 MFMA tiles)
    constexpr bool CANQ = true;
    auto k = (CANQ && args.cpg <= 4 && !bf_tune().no_quad) ? bf_nuts_pipe_kernel<W, TR, DEC, CANQ ? 1 : 0>
             : ((CANQ && args.cpg <= 8 && !bf_tune().no_quad) ? bf_nuts_pipe_kernel<W, TR, DEC, CANQ ? 2 : 0> : bf_nuts_pipe_kernel<W, TR, DEC>);
    if ((args.cpg <= 4 && !bf_tune().no_quad && bf_tune().wave_cpg == 0) || bf_tune().lone == 2) {
        const int r = launch_lone<W, TR, DEC>(ctx, args, args.n_chain, false);
        if (r != 0) return r < 0 ? r : 0;
    }
    const size_t lds = PipeGeo<W, DEC>::lds_doubles() * sizeof(double);
    if (lds > 64 * 1024)
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int groups = (args.n_chain + args.cpg - 1) / args.cpg;
    const bool two_parts = bf_tune().tail_relaunch && args.cpg == 16 && !bf_tune().no_quad && !args.stamps;
    if (two_parts) {
        if (ctx->tail_cap < args.n_chain + 2) {
            BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (ctx->tail_buf) BF_HIP_CHECK(hipFree(ctx->tail_buf));
            ctx->tail_buf = NULL;
            ctx->tail_cap = 0;
            BF_HIP_CHECK(hipMalloc((void **)&ctx->tail_buf, (size_t)(args.n_chain + 2) * sizeof(int)));
            ctx->tail_cap = args.n_chain + 2;
        }
        args.tail_stop = tail_stop_now();
        args.tail_q = bf_tune().tail_q < 1 ? 1 : (bf_tune().tail_q > 4 ? 4 : bf_tune().tail_q);
        args.tail_done = ctx->tail_buf + 1;
        BF_HIP_CHECK(hipMemsetAsync(ctx->tail_buf, 0, 2 * sizeof(int), ctx->stream));
    }
    hipLaunchKernelGGL(k, dim3(groups), dim3(1024), lds, ctx->stream, ctx->model, args);
    BF_HIP_CHECK(hipGetLastError());
    snprintf(bf_tune().last_kernel, sizeof(bf_tune().last_kernel), "bf_nuts_pipe_kernel<%d, %s, %d, %d>", W, TR ? "true" : "false", DEC,
             (args.cpg <= 4 && !bf_tune().no_quad) ? 1 : ((args.cpg <= 8 && !bf_tune().no_quad) ? 2 : 0));
    if (two_parts) {
        hipLaunchKernelGGL(bf_tail_list_kernel, dim3((args.n_chain + 255) / 256), dim3(256), 0, ctx->stream, args.n_chain, args.iter_end, args.sc,
                           ctx->tail_buf);
        BF_HIP_CHECK(hipGetLastError());
        auto k2 = bf_nuts_pipe_kernel<W, TR, DEC, 1>;
        if (lds > 64 * 1024)
            BF_HIP_CHECK(hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SamplerArgs a2 = args;
        a2.tail_stop = 0;
        a2.tail_count = ctx->tail_buf;
        a2.tail_list = ctx->tail_buf + 2;
        a2.tail_done = NULL;
        a2.cpg = 4;
        {   // the stragglers one per workgroup in the latency kernel (at most tail_stop chains of every first-part workgroup are listed)
            const int most = args.n_chain < args.tail_stop * groups ? args.n_chain : args.tail_stop * groups;
            const int r = launch_lone<W, TR, DEC>(ctx, a2, most, true);
            if (r != 0) return r < 0 ? r : 0;
        }
        // (at most four chains of every first-part workgroup are listed: groups workgroups of four chains, or one chain per CU)
        const int groups2 = groups > ctx->n_cu ? groups : ctx->n_cu;
        hipLaunchKernelGGL(k2, dim3(groups2), dim3(1024), lds, ctx->stream, ctx->model, a2);
        BF_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// test / tuning hook: keep NUTS / HMC on the common surrogate off the group kernel (bfhip_group.hip), i.e. on the
// kernels of this file (also selected by BFHIP_NUTS_KERNEL=sliced or =pipe)


// diagnostics hook (not part of include/bfhip.h): per-wave cycle counters of the sampler kernel's phases

// test / tuning hook (not part of include/bfhip.h; also BFHIP_PLD_WAVES): 8 or 16 waves per workgroup for the pipeline density, 0 = by chain count


template <int W, bool NUTS>
static int launch_sampler(bfhip_ctx *ctx, const SamplerArgs &args) {
    const DevModel &m = ctx->model;
    const bool plain = sampler_plain(m) && !args.mat;
    if (m.pld.on) {   // pipeline density: the FS = 8 / 9 / 10 instantiations
        if constexpr (W == 8) {
            // d = 128 (round 6): the eight-wave form with the run-time feature set, two dimensions per lane in phase P
            if (sampler_lds_bytes(m, false, 8) > (size_t)160 * 1024)
                return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_sampler_run: this pipeline density needs %zu KB of LDS at d = %d (160 KB)",
                                    sampler_lds_bytes(m, false, 8) / 1024, m.d);
            return args.mat ? launch_sampler_t<8, NUTS, false, 9, 1>(ctx, args) : launch_sampler_t<8, NUTS, false, 9>(ctx, args);
        }
        constexpr int WP = W <= 4 ? W : 1;   // (keeps W = 8 from instantiating the other forms)
        if (args.mat) {   // full-rank metric: the eight-wave form with the run-time feature set (bfhip_metric.h streams the chain's own matrices)
            return launch_sampler_t<WP, NUTS, false, 9, 1>(ctx, args);
        }
        // eight chains per workgroup (and 256 registers a wave) while that fills the chip, sixteen beyond
        // (measured, tools/pld_rate.py: with the outputs compressed to the monomial count the DES shape's contractions are 200
        // tile k-steps and the eight-wave form wins at every chain count -- 1.41 against 1.30 x 10^8 at 4096 chains; at 1800 tile
        // k-steps, a quadratic config on 20 inputs, the sixteen-wave form's full tiles win, 8.2 against 7.3 x 10^7)
        const long gemm_steps = (long)m.pld.NT1 * m.pld.NS1 + (long)m.pld.NT2 * m.pld.NS2;
        const bool w8 = m.pld.only8 || (bf_tune().pld_waves ? bf_tune().pld_waves == 8 : (args.n_chain <= 8 * ctx->n_cu || gemm_steps <= 800));
        if (w8 && m.has_transform && m.has_su && m.use_bound && !m.use_decay && !(bf_tune().no_plain != 0)) return launch_sampler_t<WP, NUTS, false, 10>(ctx, args);
        return w8 ? launch_sampler_t<WP, NUTS, false, 9>(ctx, args) : launch_sampler_t<WP, NUTS, false, 8>(ctx, args);
    }
    if (args.mat) {   // (a compile-time feature set changes nothing here: 7.4 x 10^7 either way)
        // While the metric adapts an iteration ends with passes over three more matrices, and the second column of the
        // velocity-ahead pass costs more than the pass it saves (3.0 against 3.3 x 10^7, tools/full_metric_rate.py); afterwards it
        // is worth a fifth (7.4 -> 8.9 x 10^7).  Both forms in one kernel were slower than either.  The same numbers from both.
        const bool adapting = args.cfg.adapt_metric && args.iter_out0 < args.cfg.n_warmup;
        return (adapting || (bf_tune().no_vel_ahead != 0)) ? launch_sampler_t<W, NUTS, false, 0, 1>(ctx, args) : launch_sampler_t<W, NUTS, false, 0, 2>(ctx, args);
    }
#ifndef BF_TRACE
    if (W == 4 && NUTS && args.stamps)  // diagnostic build, d <= 64 NUTS only
        return plain ? launch_sampler_t<W, NUTS, (W == 4 && NUTS), (W == 4 && NUTS) ? 1 : 0>(ctx, args)
                     : launch_sampler_t<W, NUTS, (W == 4 && NUTS), 0>(ctx, args);
#endif
#ifdef BF_TRACE   // (tuning builds: the pipelined kernel writes its own stamps)
    const bool stamped = false;
#else
    const bool stamped = args.stamps != NULL;
#endif
    if (plain && NUTS && W <= 4 && !(bf_tune().no_pipe != 0) && !stamped)
        return launch_nuts_pipe<(W <= 4 ? W : 1)>(ctx, args);
    if (plain) return launch_sampler_t<W, NUTS, false, 1>(ctx, args);
    // ... and the same surrogate behind the constraint transform (bounded parameters)
    if (W <= 4 && NUTS && !(bf_tune().no_pipe != 0) && !(bf_tune().no_plain != 0) && !stamped && m.has_quad && m.use_bound && m.has_transform && !m.use_decay &&
        !m.has_su && !m.has_cubic && !m.has_link)
        return launch_nuts_pipe<(W <= 4 ? W : 1), (W <= 4)>(ctx, args);
    // ... and with the decay penalty (the GBS recipes' densities: configs 3 and 4)
    if (W <= 4 && NUTS && !(bf_tune().no_pipe != 0) && !(bf_tune().no_plain != 0) && !stamped && m.has_quad && m.use_bound && m.use_decay &&
        !m.has_transform && !m.has_su && !m.has_cubic && !m.has_link) {
        // (the decay term's matrix and centre are the bound's, bit for bit: two matrices do, bfhip_nuts_pipe.h)
        if (m.decay_shared && !bf_tune().no_decay_shared) return launch_nuts_pipe<(W <= 4 ? W : 1), false, (W <= 4 ? 2 : 0)>(ctx, args);
        return launch_nuts_pipe<(W <= 4 ? W : 1), false, (W <= 4 ? 1 : 0)>(ctx, args);
    }
#ifndef BF_ONLY_HEADLINE
    // the common surrogate with the decay penalty and / or the constraint transform: compile-time feature sets at
    // 33 <= d <= 64 (the optional features' branches and register arrays of the run-time kernel disappear)
    if (W == 4 && !(bf_tune().no_plain != 0) && m.has_quad && m.use_bound && !m.has_su && !m.has_cubic && !m.has_link) {
        constexpr int W4 = W == 4 ? 4 : W;  // (keeps the other W from instantiating these)
        if (m.use_decay && m.has_transform) return launch_sampler_t<W4, NUTS, false, (W == 4 ? 7 : 0)>(ctx, args);
        if (m.use_decay) return launch_sampler_t<W4, NUTS, false, (W == 4 ? 3 : 0)>(ctx, args);
        if (m.has_transform) return launch_sampler_t<W4, NUTS, false, (W == 4 ? 5 : 0)>(ctx, args);
    }
    // d = 128 with cubic configs and nothing else (config 5): the feature set fixed at compile time too
    if (W == 8 && NUTS && !(bf_tune().no_plain != 0) && m.has_quad && m.use_bound && m.has_cubic && !m.use_decay && !m.has_transform && !m.has_su &&
        !m.has_link) {
        // at most four chains per CU (config 5's shard): four waves, each with a chain and two row tiles of S in registers
        const int form = bf_tune().cubic_form;
        if (form != 8 && bf_tune().wave_cpg == 0 && sampler_ksplit(m) == 1 && sampler_cubic_lds(m, false) &&
            (form == 4 || args.n_chain <= 4 * ctx->n_cu))
            return launch_sampler_t<(W == 8 ? 8 : W), NUTS, false, (W == 8 && NUTS ? 17 : 0)>(ctx, args);
        return launch_sampler_t<(W == 8 ? 8 : W), NUTS, false, (W == 8 && NUTS ? 16 : 0)>(ctx, args);
    }
#endif
    return launch_sampler_t<W, NUTS, false, 0>(ctx, args);
}

extern "C" int bfhip_sampler_run(bfhip_ctx *ctx, const bfhip_sampler_config *cfg, int n_chain, int iter_end,
                                 uint64_t *rng, double *sc, double *vec, int iter_out0, int n_out, double *samples,
                                 double *stats, unsigned long long *n_leapfrog) {
    BfDeviceGuard dev_guard(ctx);
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
    args.cpg = 0;
    args.cub_lds = 0;
    args.no_quad = bf_tune().no_quad;
    args.cfg = *cfg;
    args.n_chain = n_chain;
    args.iter_end = iter_end;
    args.iter_out0 = iter_out0;
    args.n_out = n_out;
    args.nslot = SL_PIPE_N;  // both ends, proposal, p_sum, 4 vectors per stack level (+ the proposals' gradients: pipelined kernel)
    args.tail_max = bf_tune().tail_max;
    args.ks = sampler_ksplit(m);
    args.gbn = sampler_gb_slots(m);
    args.rng = rng;
    args.sc = sc;
    args.vec = vec;
    args.samples = samples;
    args.stats = stats;
    args.n_leapfrog = n_leapfrog;
    args.stamps = bf_tune().stamps;
    args.no_bound_proof = bf_no_bound_proof();
    args.gcount = NULL;
    args.mat = (cfg->full_metric && cfg->metric_mat) ? cfg->metric_mat : NULL;
    if (cfg->full_metric && !cfg->metric_mat) return bf_set_error(BFHIP_ERR_ARG, "bfhip_sampler_run: full_metric without metric_mat");
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
    // the common surrogate (linear + quadratic configs with the bound; decay and constraint transform optional) at
    // d <= 64 with the diagonal metric: the group kernel
    if (cfg->chain_layout < 0 || cfg->chain_layout > 3) return bf_set_error(BFHIP_ERR_ARG, "chain_layout should be 0, 1, 2 or 3");
    if (cfg->chain_layout == 3 && !(bf_tune().no_group != 0) && !(bf_tune().no_pipe != 0) && !(bf_tune().no_plain != 0) && !args.stamps && bf_split_supports(m, args)) {
        snprintf(bf_tune().last_kernel, sizeof(bf_tune().last_kernel), "bf_split_kernel<%d>", W);
        return bf_launch_split(ctx, args);
    }
    const bool want_group = cfg->chain_layout == 1 || cfg->chain_layout == 3 || (cfg->chain_layout == 0 && !nuts);
    if (want_group && !(bf_tune().no_group != 0) && !(bf_tune().no_pipe != 0) && !(bf_tune().no_plain != 0) && !args.stamps && bf_group_supports(m, args)) {
        snprintf(bf_tune().last_kernel, sizeof(bf_tune().last_kernel), "bf_group_kernel<%d, %s, %d>", W, nuts ? "true" : "false",
                 m.pld.on ? (8 | (m.has_transform ? 4 : 0)) : (1 | (m.use_decay ? 2 : 0) | (m.has_transform ? 4 : 0)));
        return bf_launch_group(ctx, args);
    }
    {
        // (the conditions of launch_sampler: the common surrogate, plain or behind the constraint transform)
        const bool common = m.has_quad && m.use_bound && !m.has_su && !m.has_cubic && !m.has_link && !(bf_tune().no_plain != 0);
        const bool tr_only = common && m.has_transform && !m.use_decay, dec_only = common && m.use_decay && !m.has_transform;
        const bool pipe = nuts && W <= 4 && !(bf_tune().no_pipe != 0) && !args.mat && !args.stamps && (sampler_plain(m) || tr_only || dec_only);
        if (m.pld.on) snprintf(bf_tune().last_kernel, sizeof(bf_tune().last_kernel), "bf_sampler_kernel<%d, %s, false, 8 | 9>", W, nuts ? "true" : "false");
        else snprintf(bf_tune().last_kernel, sizeof(bf_tune().last_kernel), "%s<%d, ...>", pipe ? "bf_nuts_pipe_kernel" : "bf_sampler_kernel", W);
    }
    switch (W) {
#ifndef BF_ONLY_HEADLINE  // tuning builds (-DBF_ONLY_HEADLINE) compile the 64-d instantiations only
    case 1: return nuts ? launch_sampler<1, true>(ctx, args) : launch_sampler<1, false>(ctx, args);
    case 2: return nuts ? launch_sampler<2, true>(ctx, args) : launch_sampler<2, false>(ctx, args);
    case 8: return nuts ? launch_sampler<8, true>(ctx, args) : launch_sampler<8, false>(ctx, args);
#endif
    case 4: return nuts ? launch_sampler<4, true>(ctx, args) : launch_sampler<4, false>(ctx, args);
    }
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "unsupported padded dimension %d", m.DP);
}


// ---- full-rank metric initialisation (one wave per chain) ------------------------------------------
template <int E>
__global__ __launch_bounds__(64) void bf_metric_init_full_kernel(int n_chain, int d, const double *cov0, double initial_weight,
                                                               double *sc, double *mat) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    if (chain >= n_chain) return;
    const size_t msz = (size_t)d * d;
    double *mp = mat + (size_t)chain * BF_MAT_N * msz;
    double *covT = mp + BF_MAT_COV * msz, *fgT = mp + BF_MAT_FG * msz, *bgT = mp + BF_MAT_BG * msz, *wT = mp + BF_MAT_WORK * msz;
    for (int j = 0; j < d; ++j) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            if (i < d) {
                const double c = cov0 ? cov0[(size_t)i * d + j] : (i == j ? 1. : 0.);  // cov[i][j] -> covT[j][i]
                covT[(size_t)j * d + i] = c;
                fgT[(size_t)j * d + i] = c * initial_weight;   // _WeightedCovariance(n, mean, cov, weight): metrics.py:382-395
                bgT[(size_t)j * d + i] = (i == j) ? 10. : 0.;  // _WeightedCovariance(n)
            }
        }
    }
    if (bf_chol_rows<E>(covT, wT, d, lane)) {
        bf_chol_publish<E>(wT, mp + BF_MAT_CHOL * msz, mp + BF_MAT_CHOL_ROWS * msz, d, lane);
    } else if (lane == 0) {
        sc[(size_t)chain * BFHIP_SC_N + BFHIP_SC_ERROR] = 3.;
    }
}

extern "C" int bfhip_metric_init_full(bfhip_ctx *ctx, int n_chain, int d, const double *cov0, double initial_weight,
                                      double *sc, double *mat) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n_chain < 0 || d <= 0 || d > BFHIP_MAX_DIM || (n_chain > 0 && (!sc || !mat)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_metric_init_full: invalid argument");
    if (n_chain == 0) return 0;
    if (d <= 64)
        hipLaunchKernelGGL(bf_metric_init_full_kernel<1>, dim3(n_chain), dim3(64), 0, ctx->stream, n_chain, d, cov0, initial_weight, sc, mat);
    else
        hipLaunchKernelGGL(bf_metric_init_full_kernel<2>, dim3(n_chain), dim3(64), 0, ctx->stream, n_chain, d, cov0, initial_weight, sc, mat);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// "Did the chains run in step?" -- the share of the most common NUTS tree size among rows [row0, row0 + n_rows) of the
// statistics of all chains, compared with a threshold -- and "does some chain lag far behind?": a launch lasts as long as its
// busiest chain, so the largest per-chain sum of tree sizes over the window is compared with the mean.  work
// (BFHIP_TREE_MODE_WORK int32, zeroed once by the caller): [0] the answer, [1 .. 4096] histogram of the sizes 0 .. 4095,
// [4097] arrival counter, [4098 .. 4098 + 63] chains by the size class of their window sum (bf_lag_class); every workgroup adds
// its part, the last one to arrive decides and clears the buffer for the next call.  Integers only, and sums that do not depend
// on how the chains are split over launches or ranks: the sharded path (chains.py: hist_reduce) takes the same decision from the
// ranks' summed histograms.
// ---------------------------------------------------------------------------------------------------
// size classes 1, 2, 3, 4, 6, 8, 12, 16, ... (lower edges; class j holds the sums in [edge j, edge j + 1))
__host__ __device__ inline long bf_lag_edge(int j) { return j < 2 ? j + 1 : ((j & 1) ? 1L << ((j + 1) / 2) : 3L << (j / 2 - 1)); }
__host__ __device__ inline int bf_lag_class(long v) {
    int j = 0;
    while (j < 63 && bf_lag_edge(j + 1) <= v) ++j;
    return j;
}
__global__ __launch_bounds__(256) void bf_tree_mode_kernel(int n_chain, int n_out, const double *__restrict__ stats, int row0,
                                                           int n_rows, double share, int *__restrict__ work) {
    __shared__ unsigned int hist[4096];
    __shared__ unsigned int lagc[64];
    __shared__ unsigned int best;
    __shared__ int last;
    __shared__ unsigned int mode_at;
    __shared__ unsigned long long tot_sum;
    for (int i = threadIdx.x; i < 4096; i += 256) hist[i] = 0;
    if (threadIdx.x < 64) lagc[threadIdx.x] = 0;
    if (threadIdx.x == 0) { best = 0; mode_at = 0xFFFFFFFFu; tot_sum = 0; }
    __syncthreads();
    const long total = (long)n_chain * n_rows;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long c = i / n_rows, r = row0 + i % n_rows;
        const double t = stats[(c * n_out + r) * BFHIP_STAT_STRIDE + BFHIP_NS_TREE_SIZE];
        const int b = t >= 0. && t < 4095. ? (int)t : 4095;
        atomicAdd(&hist[b], 1u);
    }
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < n_chain; c += (long)gridDim.x * 256) {   // this chain's leapfrogs in the window
        long sum = 0;
        for (int r = 0; r < n_rows; ++r) {
            const double t = stats[(c * n_out + row0 + r) * BFHIP_STAT_STRIDE + BFHIP_NS_TREE_SIZE];
            sum += t >= 0. && t < 4095. ? (int)t : 4095;
        }
        atomicAdd(&lagc[bf_lag_class(sum)], 1u);
    }
    __syncthreads();
    unsigned int *gh = (unsigned int *)work + 1, *gl = (unsigned int *)work + 4098;
    for (int i = threadIdx.x; i < 4096; i += 256)
        if (hist[i]) atomicAdd(&gh[i], hist[i]);
    if (threadIdx.x < 64 && lagc[threadIdx.x]) atomicAdd(&gl[threadIdx.x], lagc[threadIdx.x]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&work[4097], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    unsigned int mx = 0, at = 0;
    unsigned long long part = 0;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const unsigned int v = atomicExch(&gh[i], 0u);  // (read through the atomic path, and cleared for the next call)
        part += (unsigned long long)v * (unsigned)i;
        if (v > mx) { mx = v; at = (unsigned)i; }
    }
    if (threadIdx.x < 64) lagc[threadIdx.x] = atomicExch(&gl[threadIdx.x], 0u);
    atomicMax(&best, mx);
    atomicAdd(&tot_sum, part);
    __syncthreads();
    if (mx == best && mx > 0) atomicMin(&mode_at, at);   // (the smallest size among equally common ones)
    __syncthreads();
    if (threadIdx.x == 0) {
        const int mode = mode_at < 4096u ? (mode_at > 1u ? (int)mode_at : 1) : 1;
        int top = 0;
        for (int j = 0; j < 64; ++j)
            if (lagc[j]) top = j;
        // (the busiest chain's window sum is at least its class's lower edge: at least FOUR times the mean, in integers -- at twice the
        // mean the wave layout's second part only breaks even at 64-d, and one chain in thousands at 2x is likely while the step sizes
        // still adapt; the measurement behind the rule is a 16x chain)
        const bool lag = (unsigned long long)bf_lag_edge(top) * (unsigned long long)n_chain >= 4ull * tot_sum && tot_sum > 0;
        // [0] the common size when the trees are in step, else 0; + 4096: some chain lags far behind the rest (reported whether or not
        // the trees are in step: a launch lasts as long as its busiest chain either way)
        work[0] = (((double)best >= share * (double)total) ? mode : 0) | (lag ? 1 << 12 : 0);
        work[4097] = 0;
    }
}

extern "C" int bfhip_tree_size_mode_share(bfhip_ctx *ctx, int n_chain, int n_out, const double *stats, int row0, int n_rows,
                                          double share, int *work) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n_chain < 1 || n_out < 1 || !stats || row0 < 0 || n_rows < 1 || row0 + n_rows > n_out || !work)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_tree_size_mode_share: invalid argument");
    const long total = (long)n_chain * n_rows;
    int grid = (int)((total + 1023) / 1024);
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(bf_tree_mode_kernel, dim3(grid), dim3(256), 0, ctx->stream, n_chain, n_out, stats, row0, n_rows, share, work);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
