// bfhip_pld.h -- the "pipeline density": a multi-output PolyModel surrogate followed by a Gaussian likelihood of its
// outputs and an optional Gaussian prior of the inputs (core/density.py:527-560,
// modules/poly.py:430-503; examples/des-y1-w-cosmosis.ipynb cells 12-18), evaluated for the 16 chains / points of a
// workgroup WITHOUT ever forming the (m, d) Jacobian.
//
// Feature form.  Every output is a polynomial in the same nf monomials phi_p(x) = x_i1 x_i2 x_i3 (index DP = the constant 1):
//     f = C phi(x),            C (m, nf) the coefficient matrix (masks scattered, zero where a config does not reach),
//     logp = logp0 - |L^T (f - y)|^2 / 2        with prec = L L^T,
//     grad = -(d phi / d x)^T C^T prec (f - y).
// The precision's Cholesky factor is folded into the coefficients when the density is uploaded (C' = L^T C, y' = L^T y,
// f_mu' = L^T f_mu: the bound's extrapolation, modules/poly.py:480-503, is linear in f_0 and f_mu, so it commutes with the
// whitening), which removes the m x m product from the kernel; when there are more outputs than monomials the output space is
// compressed to nf rows by a Householder factorisation C' = Q [R; 0], exactly (bfhip_pld.hip): what is left are two dense
// contractions on the FP64 matrix cores with the 16 chains as the 16 columns,
//     GEMM1   F (MP x 16)  = C' (MP x PP)   Phi (PP x 16)         r = F - y'  (bound: F extrapolated first)
//     GEMM2   W (PP x 16)  = C'^T (PP x MP) R   (MP x 16)
// and a sparse per-chain contraction  (J_0^T r)_j = sum_e mult_e W[p_e] x[a_e] x[b_e]  over the monomials that contain x_j.
// C' and C'^T are kept as MFMA A-operand fragments in global memory (they stay in L2: 2 x 267 KB at the DES shape
// m = 457, nf = 73) and streamed once per trip by the workgroup; Phi, R and W live in LDS in B-operand layout.
//
//   A fragments   CF [(t * NS1 + s) * 64 + l] = C'[16 t + (l & 15)][4 s + (l >> 4)]      t < NT1 = MP / 16, s < NS1 = PP / 4
//                 CTF[(u * NS2 + s) * 64 + l] = C'[4 s + (l >> 4)][16 u + (l & 15)]      u < NT2 = PP / 16, s < NS2 = MP / 4
//   B operands    X[(k >> 2) * XS + c + 16 * (k & 3)] = value of row k for chain c  (PLD_XS = 65: conflict-free for the
//                 chain waves' column writes; a k-step is one contiguous 64-lane read)
#pragma once
#include "bfhip_model.h"

#define PLD_XS 65
#define PLD_MAX_KS2 8

// B-operand rows of the eight-chain form: [k = 0 .. 3][chain 0 .. 7] + 1 (the sixteen-chain forms: [k][chain 0 .. 15] + 1 = PLD_XS).
// Half the LDS per monomial and per output row: what lets a surrogate of more than ~350 monomials (the 27-d full quadratic of the
// DES example: 406) run at all (bfhip_pipeline_upload: only8).
#define PLD_XS8 33

// LDS regions of the pipeline block (doubles), in this order behind `base`
struct PldLds {
    int XS, CW;    // row stride of the B operands and chains per k of a row: 65 / 16, or 33 / 8 (eight-chain form)
    double *XE;    // [16][DP + 2]  evaluation point of every chain, then 1 and 0
    double *CH;    // [16]          beta of the chains outside the bound's ellipsoid, 0 inside
    double *YW;    // [2][MP]       y' and f_mu'
    double *RED;   // [16][2][16]   per wave and chain: sum r^2, sum (f_0 - f_mu) r over the wave's row tiles
    double *RB;    // [NS2][XS]     r as the B operand of GEMM2
    double *PHI;   // [NS1][XS]     monomials as the B operand of GEMM1; slot 0 of W afterwards
    double *WX;    // [KS2 - 1][NS1][XS]  further partial-sum slots of W
    const unsigned long long *GT;   // [n_ent][DP]  gradient table (staged from pl.gtab once per launch)
    const unsigned *MONO;           // [PP]         monomial table (staged from pl.mono)
    double *CL;    // [MP][PP + 1] row-major copy of C' (eight-chain forms, when it fits: both contractions read their A operands from it
                   //              instead of streaming the fragments from L2 in every trip), or NULL
    int CLS;       // its row stride (odd: the 16 rows a tile's lanes read fall into 16 different bank pairs)
};

__host__ __device__ inline size_t pld_cl_doubles(int MP, int PP) { return (size_t)MP * (PP + 1) + 1; }

// (n_red: waves that post partial sums into RED, 16 unless the caller has fewer -- the group kernel's two or four)
__host__ __device__ inline size_t pld_lds_doubles(int DP, int MP, int PP, int KS2, int n_ent, int xs = PLD_XS, int n_red = 16) {
    const size_t ns1 = PP / 4, ns2 = MP / 4;
    return (size_t)16 * (DP + 2) + 16 + (size_t)2 * MP + (size_t)32 * n_red + ns2 * xs + (size_t)KS2 * ns1 * xs + (size_t)n_ent * DP + PP / 2;
}

#ifndef BF_HOST_EMU
__device__ inline PldLds pld_lds(double *base, int DP, const PldDev &pl, int cw = 16, bool with_cl = false, int n_red = 16) {
    PldLds L;
    L.CW = cw;
    L.XS = cw == 8 ? PLD_XS8 : PLD_XS;
    L.XE = base;
    L.CH = L.XE + 16 * (DP + 2);
    L.YW = L.CH + 16;
    L.RED = L.YW + 2 * pl.MP;
    L.RB = L.RED + 32 * n_red;
    L.PHI = L.RB + (size_t)pl.NS2 * L.XS;
    L.WX = L.PHI + (size_t)pl.NS1 * L.XS;
    double *gt = L.WX + (size_t)(pl.KS2 - 1) * pl.NS1 * L.XS;
    L.GT = (const unsigned long long *)gt;
    L.MONO = (const unsigned *)(gt + (size_t)pl.n_ent * DP);
    L.CLS = pl.PP + 1;
    L.CL = with_cl ? gt + (size_t)pl.n_ent * DP + ((pl.PP / 2 + 1) & ~1) : nullptr;   // (behind the monomial words, 16-byte aligned)
    return L;
}

// once per launch, all threads: y', f_mu' and the two index tables into LDS
__device__ inline void pld_stage(const PldDev &pl, const PldLds &L, int DP, int tid, int nth) {
    for (int i = tid; i < pl.MP; i += nth) {
        L.YW[i] = pl.yw[i];
        L.YW[pl.MP + i] = pl.fmuw[i];
    }
    unsigned long long *gt = (unsigned long long *)L.GT;
    for (int i = tid; i < pl.n_ent * DP; i += nth) gt[i] = pl.gtab[i];
    unsigned *mo = (unsigned *)L.MONO;
    for (int i = tid; i < pl.PP; i += nth) mo[i] = pl.mono[i];
    if (L.CL) {   // CF[(t NS1 + s) 64 + l] = C'[16 t + (l & 15)][4 s + (l >> 4)]  ->  CL[row][col]
        const int n = pl.NT1 * pl.NS1 * 64;
        for (int i = tid; i < n; i += nth) {
            const int l = i & 63, s2 = (i >> 6) % pl.NS1, t = (i >> 6) / pl.NS1;
            L.CL[(size_t)(16 * t + (l & 15)) * L.CLS + 4 * s2 + (l >> 4)] = pl.CF[i];
        }
    }
}

// chain wave c (lane = dimension): the evaluation point, beta (0 inside the bound) and the monomials of the chain
__device__ inline void pld_point(const PldDev &pl, const PldLds &L, int DP, int c, int lane, double x_eval, double beta_oob) {
    double *xe = L.XE + c * (DP + 2);
    if (lane < DP) xe[lane] = x_eval;
    if (lane == 0) {
        xe[DP] = 1.;
        xe[DP + 1] = 0.;
        L.CH[c] = beta_oob;
    }
    // (the wave's own LDS writes are visible to its later reads: one wave's LDS operations complete in order)
    for (int p0 = 0; p0 < pl.PP; p0 += 64) {
        const int p = p0 + lane;
        if (p < pl.PP) {
            const unsigned mo = L.MONO[p];
            const double v = (xe[mo & 255u] * xe[(mo >> 8) & 255u]) * xe[(mo >> 16) & 255u];
            L.PHI[(p >> 2) * L.XS + c + L.CW * (p & 3)] = v;
        }
    }
}

// the same for a wave that holds E dimensions per lane (dimension lane E + e: d = 128, E = 2)
template <int E>
__device__ inline void pld_point_e(const PldDev &pl, const PldLds &L, int DP, int c, int lane, const double (&x_eval)[E], double beta_oob) {
    double *xe = L.XE + c * (DP + 2);
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (lane * E + e < DP) xe[lane * E + e] = x_eval[e];
    if (lane == 0) {
        xe[DP] = 1.;
        xe[DP + 1] = 0.;
        L.CH[c] = beta_oob;
    }
    for (int p0 = 0; p0 < pl.PP; p0 += 64) {
        const int p = p0 + lane;
        if (p < pl.PP) {
            const unsigned mo = L.MONO[p];
            const double v = (xe[mo & 255u] * xe[(mo >> 8) & 255u]) * xe[(mo >> 16) & 255u];
            L.PHI[(p >> 2) * L.XS + c + L.CW * (p & 3)] = v;
        }
    }
}

__device__ inline double pld_rowsum4(double v) {   // sum over the four 16-lane rows of the wave, in every lane
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    const int lo2 = __double2loint(v), hi2 = __double2hiint(v);
    const auto a2 = __builtin_amdgcn_permlane32_swap(lo2, lo2, false, false);
    const auto b2 = __builtin_amdgcn_permlane32_swap(hi2, hi2, false, false);
    return __hiloint2double(b2[0], a2[0]) + __hiloint2double(b2[1], a2[1]);
}

// one (row tile, K range) contraction: acc += A[tile][s0 .. s1) B[s0 .. s1); A fragments from global memory (L2), B from LDS.
// Four k-steps are fetched while the four before them run on the matrix pipe.
__device__ inline d4_t pld_tile(const double *__restrict__ Af, const double *Bf, int n_steps, int lane) {
    d4_t acc = {0., 0., 0., 0.};
    const double *ap = Af + lane;
    const double *bp = Bf + lane;
    double a0[4], b0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a0[q] = ap[q * 64]; b0[q] = bp[q * PLD_XS]; }
    for (int s = 4; s < n_steps; s += 4) {   // (n_steps is a multiple of 4)
        double a1[4], b1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a1[q] = ap[(s + q) * 64]; b1[q] = bp[(s + q) * PLD_XS]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { a0[q] = a1[q]; b0[q] = b1[q]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], acc, 0, 0, 0);
    return acc;
}

// two row tiles against the SAME B operand (GEMM1: both tiles multiply Phi): two independent accumulation chains per wave, one
// LDS read per k-step for both
__device__ inline void pld_tile2(const double *__restrict__ Af0, const double *__restrict__ Af1, const double *Bf, int n_steps, int lane,
                                 d4_t &acc0, d4_t &acc1) {
    acc0 = d4_t{0., 0., 0., 0.};
    acc1 = d4_t{0., 0., 0., 0.};
    const double *ap0 = Af0 + lane, *ap1 = Af1 + lane, *bp = Bf + lane;
    double x0[4], y0[4], b0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { x0[q] = ap0[q * 64]; y0[q] = ap1[q * 64]; b0[q] = bp[q * PLD_XS]; }
    for (int s = 4; s < n_steps; s += 4) {
        double x1[4], y1[4], b1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { x1[q] = ap0[(s + q) * 64]; y1[q] = ap1[(s + q) * 64]; b1[q] = bp[(s + q) * PLD_XS]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], b0[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[q], b0[q], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { x0[q] = x1[q]; y0[q] = y1[q]; b0[q] = b1[q]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], b0[q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[q], b0[q], acc1, 0, 0, 0);
    }
}

// the epilogue of one row tile of GEMM1: the bound's extrapolation per chain, r = F - y' into the B operand of GEMM2, and the
// lane's contributions to sum r^2 and sum (f_0 - f_mu) r (reduced once per wave, after its last tile: pld_red_put)
__device__ inline void pld_epilogue1(const PldDev &pl, const PldLds &L, double alpha, double inv_alpha, double beta, int t, const d4_t &acc,
                                     int lane, double &s_rr, double &s_fr) {
    const int mg = lane >> 4;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int row = 16 * t + 4 * r4 + mg;
        const double f0 = acc[r4], y = L.YW[row], fmu = L.YW[pl.MP + row];
        const double fv = beta > 0. ? (beta * f0 - (beta - alpha) * fmu) * inv_alpha : f0;   // modules/poly.py:487
        const double r = fv - y;
        L.RB[(4 * t + r4) * PLD_XS + lane] = r;   // row >> 2 = 4 t + r4, 16 (row & 3) + chain = lane
        s_rr += r * r;
        s_fr += (f0 - fmu) * r;
    }
}
// the wave's two sums per chain (lane & 15 = chain; the four 16-lane rows hold different output rows) into its RED slot
__device__ inline void pld_red_put(const PldLds &L, int w, int lane, double s_rr, double s_fr) {
    s_rr = pld_rowsum4(s_rr);
    s_fr = pld_rowsum4(s_fr);
    if (lane < 16) {
        L.RED[(w * 2 + 0) * 16 + lane] = s_rr;
        L.RED[(w * 2 + 1) * 16 + lane] = s_fr;
    }
}

// GEMM1 and its epilogue, all NWV waves of the workgroup: F_0 = C' Phi, row tiles dealt two at a time
__device__ inline void pld_gemm1(const PldDev &pl, const PldLds &L, double alpha, int w, int nwv, int lane) {
    const double beta = L.CH[lane & 15], inv_alpha = 1. / alpha;
    double s_rr = 0., s_fr = 0.;
    for (int t = w; t < pl.NT1; t += 2 * nwv) {
        const int t2 = t + nwv;
        if (t2 < pl.NT1) {
            d4_t a0, a1;
            pld_tile2(pl.CF + (size_t)t * pl.NS1 * 64, pl.CF + (size_t)t2 * pl.NS1 * 64, L.PHI, pl.NS1, lane, a0, a1);
            pld_epilogue1(pl, L, alpha, inv_alpha, beta, t, a0, lane, s_rr, s_fr);
            pld_epilogue1(pl, L, alpha, inv_alpha, beta, t2, a1, lane, s_rr, s_fr);
        } else {
            const d4_t a0 = pld_tile(pl.CF + (size_t)t * pl.NS1 * 64, L.PHI, pl.NS1, lane);
            pld_epilogue1(pl, L, alpha, inv_alpha, beta, t, a0, lane, s_rr, s_fr);
        }
    }
    pld_red_put(L, w, lane, s_rr, s_fr);
}

// two independent (row tile, K range) contractions side by side (GEMM2: different A and different B): two accumulation chains
// per wave keep the matrix pipe fed when only two waves share a SIMD
__device__ inline void pld_tile2b(const double *__restrict__ Af0, const double *Bf0, const double *__restrict__ Af1, const double *Bf1,
                                  int n_steps, int lane, d4_t &acc0, d4_t &acc1) {
    acc0 = d4_t{0., 0., 0., 0.};
    acc1 = d4_t{0., 0., 0., 0.};
    const double *ap0 = Af0 + lane, *ap1 = Af1 + lane, *bp0 = Bf0 + lane, *bp1 = Bf1 + lane;
    double x0[4], y0[4], b0[4], c0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { x0[q] = ap0[q * 64]; y0[q] = ap1[q * 64]; b0[q] = bp0[q * PLD_XS]; c0[q] = bp1[q * PLD_XS]; }
    for (int s = 4; s < n_steps; s += 4) {
        double x1[4], y1[4], b1[4], c1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x1[q] = ap0[(s + q) * 64]; y1[q] = ap1[(s + q) * 64];
            b1[q] = bp0[(s + q) * PLD_XS]; c1[q] = bp1[(s + q) * PLD_XS];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], b0[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[q], c0[q], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { x0[q] = x1[q]; y0[q] = y1[q]; b0[q] = b1[q]; c0[q] = c1[q]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], b0[q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[q], c0[q], acc1, 0, 0, 0);
    }
}

// GEMM2, all waves: W = C'^T R, (row tile, K part) jobs, two at a time while both have the same number of k-steps; part kp
// lands in W slot kp (slot 0 = PHI, which GEMM1 has consumed)
__device__ inline void pld_gemm2(const PldDev &pl, const PldLds &L, int w, int nwv, int lane) {
    const int n_job = pl.NT2 * pl.KS2;
    auto steps_of = [&](int job) { const int s0 = (job % pl.KS2) * pl.KPJ2; int ns = pl.NS2 - s0; return ns > pl.KPJ2 ? pl.KPJ2 : ns; };
    auto dest = [&](int job) { const int kp = job % pl.KS2; return (kp == 0 ? L.PHI : L.WX + (size_t)(kp - 1) * pl.NS1 * PLD_XS) + (size_t)4 * (job / pl.KS2) * PLD_XS; };
    auto a_of = [&](int job) { return pl.CTF + ((size_t)(job / pl.KS2) * pl.NS2 + (job % pl.KS2) * pl.KPJ2) * 64; };
    auto b_of = [&](int job) { return L.RB + (size_t)(job % pl.KS2) * pl.KPJ2 * PLD_XS; };
    for (int job = w; job < n_job; job += 2 * nwv) {
        const int job2 = job + nwv;
        const int ns = steps_of(job);
        d4_t acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
        const bool two = job2 < n_job;
        if (two && steps_of(job2) == ns && ns > 0) {
            pld_tile2b(a_of(job), b_of(job), a_of(job2), b_of(job2), ns, lane, acc0, acc1);
        } else {
            if (ns > 0) acc0 = pld_tile(a_of(job), b_of(job), ns, lane);
            if (two && steps_of(job2) > 0) acc1 = pld_tile(a_of(job2), b_of(job2), steps_of(job2), lane);
        }
        double *W0 = dest(job);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) W0[r4 * PLD_XS + lane] = acc0[r4];
        if (two) {
            double *W1 = dest(job2);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) W1[r4 * PLD_XS + lane] = acc1[r4];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Eight-chain form (the fused sampler runs the pipeline density with eight waves of 256 registers: bfhip_sampler.hip,
// BF_PLD_WAVES): the 16-column tile would hold the FP64 pipe for 64 cycles per k-step with half of its columns empty, so a
// k-step is TWO v_mfma_f64_4x4x4_4b instead -- four 4-row blocks of the tile against the SAME four columns (chains 0-3, then
// 4-7): 35 cycles of the pipe instead of 64.  Operand maps: A lane 16 k + m as for the 16 x 16 x 4 tile; B lane 16 k + 4 b + n
// reads column n (+ 4); D lane 16 i + 4 b + n is row 4 b + i of chain n (+ 4).  Two row tiles (GEMM1) or two jobs (GEMM2) run
// side by side in a wave, eight k-steps are fetched while the eight before them run.
// ---------------------------------------------------------------------------------------------------------------------
struct PldAcc8 { double lo, hi; };   // chains 0-3 and 4-7 of one row per lane

// (ap0 / ap1: this lane's A operand of k-step 0.  ASTEP: doubles from one k-step to the next, a compile-time constant where there
// is one -- 64 for fragments in global memory, 4 for GEMM1 on the row-major LDS copy (PldLds::CL) -- so that the loads of a chunk
// keep immediate offsets; 0: run time (a_step: GEMM2 on the copy walks rows, 4 CLS doubles apart).  The chunks are fetched in
// order, so the running pointers advance by one chunk per fetch.)
template <bool SAME_B, int ASTEP>
__device__ inline void pld_tile2_q8(const double *__restrict__ ap0, const double *Bf0, const double *__restrict__ ap1, const double *Bf1,
                                    int n_steps, int lane, PldAcc8 &acc0, PldAcc8 &acc1, int XS, int CW, int a_step = 0) {
    // chunks of four k-steps (n_steps is a multiple of 4), the A fragments of TWO chunks ahead on their way while one runs: no
    // guard inside a chunk, so the loads are counted exactly (vmcnt) and stay in flight across the matrix instructions
    acc0 = PldAcc8{0., 0.};
    acc1 = PldAcc8{0., 0.};
    const int bo = CW * (lane >> 4) + (lane & 3);   // column n = lane & 3 of k = lane >> 4
    const double *bp0 = Bf0 + bo, *bp1 = Bf1 + bo;
    const int n_ch = n_steps >> 2;
    const int st = ASTEP ? ASTEP : a_step;
    double xa[4], ya[4], xb[4], yb[4], xc[4], yc[4];
    const double *fa0 = ap0, *fa1 = ap1;
    auto fetch = [&](double (&x)[4], double (&y)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { x[q] = fa0[q * st]; y[q] = fa1[q * st]; }
#ifndef PLD_KNOCK_A   // (tuning: the A fragments of ONE chunk over and over -- L1 hits -- to separate load latency from the rest)
        fa0 += 4 * st;
        fa1 += 4 * st;
#endif
    };
    auto run = [&](int c, const double (&x)[4], const double (&y)[4]) {
        double bl[4], bh[4], cl[4], chh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bl[q] = bp0[(4 * c + q) * XS];
            bh[q] = bp0[(4 * c + q) * XS + 4];
            if (!SAME_B) { cl[q] = bp1[(4 * c + q) * XS]; chh[q] = bp1[(4 * c + q) * XS + 4]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0.lo = __builtin_amdgcn_mfma_f64_4x4x4f64(x[q], bl[q], acc0.lo, 0, 0, 0);
            acc0.hi = __builtin_amdgcn_mfma_f64_4x4x4f64(x[q], bh[q], acc0.hi, 0, 0, 0);
            acc1.lo = __builtin_amdgcn_mfma_f64_4x4x4f64(y[q], SAME_B ? bl[q] : cl[q], acc1.lo, 0, 0, 0);
            acc1.hi = __builtin_amdgcn_mfma_f64_4x4x4f64(y[q], SAME_B ? bh[q] : chh[q], acc1.hi, 0, 0, 0);
        }
    };
    if (n_ch <= 0) return;
    fetch(xa, ya);
    if (n_ch > 1) fetch(xb, yb);
    // (three buffers taken in turn by position in the loop body: rotating them by register moves would make every move wait
    // for the loads just issued)
    for (int c = 0; c < n_ch; c += 3) {
        if (c + 2 < n_ch) fetch(xc, yc);
        run(c, xa, ya);
        if (c + 1 < n_ch) {
            if (c + 3 < n_ch) fetch(xa, ya);
            run(c + 1, xb, yb);
        }
        if (c + 2 < n_ch) {
            if (c + 4 < n_ch) fetch(xb, yb);
            run(c + 2, xc, yc);
        }
    }
}

__device__ inline double pld_sum_b(double v) {   // sum over the four blocks b = (lane >> 2) & 3 of a 16-lane row, in every lane
    auto ror = [](double u, int ctrl) {
        const int lo = ctrl == 4 ? __builtin_amdgcn_mov_dpp(__double2loint(u), 0x124, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(__double2loint(u), 0x128, 0xf, 0xf, true);
        const int hi = ctrl == 4 ? __builtin_amdgcn_mov_dpp(__double2hiint(u), 0x124, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(__double2hiint(u), 0x128, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    };
    v += ror(v, 4);   // row_ror:4
    v += ror(v, 8);   // row_ror:8
    return v;
}

// epilogue of one row tile of GEMM1, eight-chain form: lane 16 i + 4 b + n holds row 16 t + 4 b + i of chains n and n + 4; the
// lane's contributions to the two sums are accumulated over the wave's tiles and reduced once (pld_red_put_q8)
__device__ inline void pld_epilogue1_q8(const PldDev &pl, const PldLds &L, double alpha, double inv_alpha, const double (&beta)[2], int t,
                                        const PldAcc8 &acc, int lane, double (&s_rr)[2], double (&s_fr)[2]) {
    const int i = lane >> 4, b = (lane >> 2) & 3, n = lane & 3;
    const int row = 16 * t + 4 * b + i;
    const double y = L.YW[row], fmu = L.YW[pl.MP + row];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double f0 = h ? acc.hi : acc.lo;
        const double fv = beta[h] > 0. ? (beta[h] * f0 - (beta[h] - alpha) * fmu) * inv_alpha : f0;   // modules/poly.py:487
        const double r = fv - y;
        L.RB[(4 * t + b) * L.XS + n + 4 * h + L.CW * i] = r;   // row >> 2 = 4 t + b, row & 3 = i
        s_rr[h] += r * r;
        s_fr[h] += (f0 - fmu) * r;
    }
}
__device__ inline void pld_red_put_q8(const PldLds &L, int w, int lane, const double (&s_rr)[2], const double (&s_fr)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double a = pld_rowsum4(pld_sum_b(s_rr[h])), c = pld_rowsum4(pld_sum_b(s_fr[h]));
        if (lane < 4) {
            L.RED[(w * 2 + 0) * 16 + lane + 4 * h] = a;
            L.RED[(w * 2 + 1) * 16 + lane + 4 * h] = c;
        }
    }
}

__device__ inline void pld_gemm1_q8(const PldDev &pl, const PldLds &L, double alpha, int w, int nwv, int lane) {
    const double inv_alpha = 1. / alpha;
    const double beta[2] = {L.CH[lane & 3], L.CH[(lane & 3) + 4]};
    double s_rr[2] = {0., 0.}, s_fr[2] = {0., 0.};
    for (int t = w; t < pl.NT1; t += 2 * nwv) {
        const int t2 = t + nwv < pl.NT1 ? t + nwv : t;   // (an odd tile out is computed twice side by side: same result, same time)
        PldAcc8 a0, a1;
        if (L.CL)
            pld_tile2_q8<true, 4>(L.CL + (size_t)(16 * t + (lane & 15)) * L.CLS + (lane >> 4), L.PHI,
                                  L.CL + (size_t)(16 * t2 + (lane & 15)) * L.CLS + (lane >> 4), L.PHI, pl.NS1, lane, a0, a1, L.XS, L.CW);
        else
            pld_tile2_q8<true, 64>(pl.CF + (size_t)t * pl.NS1 * 64 + lane, L.PHI, pl.CF + (size_t)t2 * pl.NS1 * 64 + lane, L.PHI, pl.NS1, lane, a0, a1,
                                   L.XS, L.CW);
        pld_epilogue1_q8(pl, L, alpha, inv_alpha, beta, t, a0, lane, s_rr, s_fr);
        if (t2 != t) pld_epilogue1_q8(pl, L, alpha, inv_alpha, beta, t2, a1, lane, s_rr, s_fr);
    }
    pld_red_put_q8(L, w, lane, s_rr, s_fr);
}

__device__ inline void pld_gemm2_q8(const PldDev &pl, const PldLds &L, int w, int nwv, int lane) {
    const int n_job = pl.NT2 * pl.KS2;
    const int i = lane >> 4, b = (lane >> 2) & 3, n = lane & 3;
    auto steps_of = [&](int job) { const int s0 = (job % pl.KS2) * pl.KPJ2; int ns = pl.NS2 - s0; return ns > pl.KPJ2 ? pl.KPJ2 : ns; };
    auto dest = [&](int job) { const int kp = job % pl.KS2; return (kp == 0 ? L.PHI : L.WX + (size_t)(kp - 1) * pl.NS1 * L.XS) + (size_t)4 * (job / pl.KS2) * L.XS; };
    // (A operand of (row tile u, k-step s): C'^T[16 u + (l & 15)][4 s + (l >> 4)] = C'[4 s + (l >> 4)][16 u + (l & 15)])
    const int a_step = L.CL ? 4 * L.CLS : 64;
    auto a_of = [&](int job) -> const double * {
        const int u = job / pl.KS2, s0 = (job % pl.KS2) * pl.KPJ2;
        if (L.CL) return L.CL + (size_t)(4 * s0 + (lane >> 4)) * L.CLS + 16 * u + (lane & 15);
        return pl.CTF + ((size_t)u * pl.NS2 + s0) * 64 + lane;
    };
    auto b_of = [&](int job) { return L.RB + (size_t)(job % pl.KS2) * pl.KPJ2 * L.XS; };
    for (int job = w; job < n_job; job += 2 * nwv) {
        const int job2 = (job + nwv < n_job && steps_of(job + nwv) == steps_of(job)) ? job + nwv : job;
        PldAcc8 a0, a1;
        if (L.CL) pld_tile2_q8<false, 0>(a_of(job), b_of(job), a_of(job2), b_of(job2), steps_of(job), lane, a0, a1, L.XS, L.CW, a_step);
        else pld_tile2_q8<false, 64>(a_of(job), b_of(job), a_of(job2), b_of(job2), steps_of(job), lane, a0, a1, L.XS, L.CW);
        double *W0 = dest(job) + b * L.XS + n + L.CW * i;   // monomial p = 16 u + 4 b + i: p >> 2 = 4 u + b, p & 3 = i
        W0[0] = a0.lo;
        W0[4] = a0.hi;
        if (job2 != job) {
            double *W1 = dest(job2) + b * L.XS + n + L.CW * i;
            W1[0] = a1.lo;
            W1[4] = a1.hi;
        } else if (job + nwv < n_job) {   // a partner with another number of k-steps (the last K part): on its own
            const int j3 = job + nwv;
            PldAcc8 c0, c1;
            if (L.CL) pld_tile2_q8<false, 0>(a_of(j3), b_of(j3), a_of(j3), b_of(j3), steps_of(j3), lane, c0, c1, L.XS, L.CW, a_step);
            else pld_tile2_q8<false, 64>(a_of(j3), b_of(j3), a_of(j3), b_of(j3), steps_of(j3), lane, c0, c1, L.XS, L.CW);
            double *W3 = dest(j3) + b * L.XS + n + L.CW * i;
            W3[0] = c0.lo;
            W3[4] = c0.hi;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Sixteen-chain contractions with the A operands from the row-major LDS copy of C' (PldLds::CL), for the lane-per-chain group
// kernel (bfhip_group.h, FS & 8): two or four waves share both contractions there, so the L2 latency of streamed fragments is not
// hidden by other waves (15 k cycles per contraction at the DES shape); from LDS a k-step is two ~100-cycle reads.  Two row tiles
// side by side per wave (two independent accumulation chains), four k-steps of operands fetched together.  K-split 1.
// ---------------------------------------------------------------------------------------------------------------------
// One row tile against its B operand over k-steps [s0, s1) (multiples of 4), as FOUR accumulation chains (k-step q of every chunk
// of four goes to chain q) added at the end -- a tile's k-steps are one dependent chain of matrix instructions otherwise, and a
// dependent v_mfma_f64_16x16x4 issues every ~200 cycles with one wave on the SIMD (7.4 k cycles for 28 k-steps,
// profiles/r06c_trace_group_pld.log) --, the operands of the next chunk on their way while one runs.
template <int ASTEP>
__device__ inline d4_t pld_tile_cl(const double *ap, int a_step, const double *bp, int s0, int s1) {
    d4_t c0 = {0., 0., 0., 0.}, c1 = {0., 0., 0., 0.}, c2 = {0., 0., 0., 0.}, c3 = {0., 0., 0., 0.};
    const int st = ASTEP ? ASTEP : a_step;
    if (s1 <= s0) return c0;
    ap += (size_t)s0 * st;
    bp += (size_t)s0 * PLD_XS;
    // two operand buffers taken in turn (no register copies between chunks: the compiler turned those into accumulator shuffles)
    double xa[4], ba[4], xb[4], bb[4];
#define PLD_FETCH(xx, bq) do { _Pragma("unroll") for (int q = 0; q < 4; ++q) { xx[q] = ap[q * st]; bq[q] = bp[q * PLD_XS]; } ap += 4 * st; bp += 4 * PLD_XS; } while (0)
#define PLD_RUN(xx, bq) do { \
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xx[0], bq[0], c0, 0, 0, 0); \
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xx[1], bq[1], c1, 0, 0, 0); \
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xx[2], bq[2], c2, 0, 0, 0); \
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(xx[3], bq[3], c3, 0, 0, 0); } while (0)
    PLD_FETCH(xa, ba);
    int s = s0;
    for (;;) {
        const bool more_b = s + 4 < s1;
        if (more_b) PLD_FETCH(xb, bb);
        PLD_RUN(xa, ba);
        s += 4;
        if (!more_b) break;
        const bool more_a = s + 4 < s1;
        if (more_a) PLD_FETCH(xa, ba);
        PLD_RUN(xb, bb);
        s += 4;
        if (!more_a) break;
    }
#undef PLD_FETCH
#undef PLD_RUN
    return (c0 + c1) + (c2 + c3);
}

// The row tiles of a contraction dealt over nwv waves so that their k-steps balance: with the triangular C' of a compressed
// output space tile t of GEMM1 has NS1 - 4 t k-steps and tile u of GEMM2 4 u + 4 -- wave w takes tiles w, 2 nwv - 1 - w,
// 2 nwv + w, ... (a boustrophedon over the tiles in order of their length).
__device__ inline bool pld_tile_mine(int t, int w, int nwv) {
    const int r = t % (2 * nwv);
    return (r < nwv ? r : 2 * nwv - 1 - r) == w;
}

__device__ inline void pld_gemm1_cl(const PldDev &pl, const PldLds &L, double alpha, int w, int nwv, int lane) {
    const double beta = L.CH[lane & 15], inv_alpha = 1. / alpha;
    double s_rr = 0., s_fr = 0.;
    for (int t = 0; t < pl.NT1; ++t) {
        if (!pld_tile_mine(t, w, nwv)) continue;
        // (triangular C': columns left of 16 t are zero in rows 16 t .. 16 t + 15)
        const int s0 = pl.tri ? (4 * t < pl.NS1 ? 4 * t : pl.NS1) : 0;
        const d4_t acc = pld_tile_cl<4>(L.CL + (size_t)(16 * t + (lane & 15)) * L.CLS + (lane >> 4), 0, L.PHI + lane, s0, pl.NS1);
        pld_epilogue1(pl, L, alpha, inv_alpha, beta, t, acc, lane, s_rr, s_fr);
    }
    pld_red_put(L, w, lane, s_rr, s_fr);
}

// W = C'^T R (K-split 1: every job is a whole row tile, its result lands in slot 0 = PHI)
__device__ inline void pld_gemm2_cl(const PldDev &pl, const PldLds &L, int w, int nwv, int lane) {
    for (int u = 0; u < pl.NT2; ++u) {
        if (!pld_tile_mine(pl.NT2 - 1 - u, w, nwv)) continue;   // (the long tiles are the last ones here)
        // (triangular C': rows below 16 u + 15 are zero in columns 16 u .. 16 u + 15)
        const int s1 = pl.tri ? (4 * u + 4 < pl.NS2 ? 4 * u + 4 : pl.NS2) : pl.NS2;
        const d4_t acc = pld_tile_cl<0>(L.CL + (size_t)(lane >> 4) * L.CLS + 16 * u + (lane & 15), 4 * L.CLS, L.RB + lane, 0, s1);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) L.PHI[(size_t)(4 * u + r4) * PLD_XS + lane] = acc[r4];
    }
}

// Sixteen waves of 128 registers (the fused sampler above ~8 chains per CU): one row tile / one job at a time per wave -- four
// waves share a SIMD, so their accumulation chains interleave on the matrix pipe, and the registers of a second chain or of a
// deeper prefetch would be spilled.
// GEMM1 and its epilogue, all NWV waves of the workgroup: F_0 = C' Phi per row tile, the bound's extrapolation per chain,
// r = F - y' into the B operand of GEMM2, and the tile's contributions to sum r^2 and sum (f_0 - f_mu) r
__device__ inline void pld_gemm1_w16(const PldDev &pl, const PldLds &L, double alpha, int w, int nwv, int lane) {
    const double beta = L.CH[lane & 15], inv_alpha = 1. / alpha;
    double s_rr = 0., s_fr = 0.;
    for (int t = w; t < pl.NT1; t += nwv) {
        const d4_t acc = pld_tile(pl.CF + (size_t)t * pl.NS1 * 64, L.PHI, pl.NS1, lane);
        pld_epilogue1(pl, L, alpha, inv_alpha, beta, t, acc, lane, s_rr, s_fr);
    }
    pld_red_put(L, w, lane, s_rr, s_fr);
}

// GEMM2, all waves: W = C'^T R, (row tile, K part) jobs; part kp lands in W slot kp (slot 0 = PHI, which GEMM1 has consumed)
__device__ inline void pld_gemm2_w16(const PldDev &pl, const PldLds &L, int w, int nwv, int lane) {
    const int n_job = pl.NT2 * pl.KS2;
    for (int job = w; job < n_job; job += nwv) {
        const int u = job / pl.KS2, kp = job % pl.KS2;
        const int s0 = kp * pl.KPJ2;
        int ns = pl.NS2 - s0;
        if (ns > pl.KPJ2) ns = pl.KPJ2;
        double *Wd = (kp == 0 ? L.PHI : L.WX + (size_t)(kp - 1) * pl.NS1 * PLD_XS);
        d4_t acc = {0., 0., 0., 0.};
        if (ns > 0) acc = pld_tile(pl.CTF + ((size_t)u * pl.NS2 + s0) * 64, L.RB + (size_t)s0 * PLD_XS, ns, lane);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) Wd[(4 * u + r4) * PLD_XS + lane] = acc[r4];
    }
}

// chain wave c after GEMM2: the chain's sums (lane v holds wave v's part; the caller reduces over the wave) ...
__device__ inline void pld_sums(const PldDev &pl, const PldLds &L, int c, int lane, int nwv, double &s_rr, double &s_fr) {
    (void)pl;
    s_rr = lane < nwv ? L.RED[(lane * 2 + 0) * 16 + c] : 0.;
    s_fr = lane < nwv ? L.RED[(lane * 2 + 1) * 16 + c] : 0.;
}

// ... and component `dim` of J_0^T r: the monomials that contain x_dim, each times its cofactor
// (round 6, measured and dropped: four entries at a time -- the table padded to a multiple of four -- to overlap the three
// dependent LDS reads of an entry: DES shape 1.46 -> 1.02 x 10^8, the register arrays cost more than the latency they hide)
__device__ inline double pld_grad(const PldDev &pl, const PldLds &L, int DP, int c, int dim) {
    const double *xe = L.XE + c * (DP + 2);
    double g = 0.;
    for (int i = 0; i < pl.n_ent; ++i) {
        const unsigned long long en = L.GT[(size_t)i * DP + dim];
        const unsigned eh = (unsigned)(en >> 32);
        const int p = (int)(unsigned)en, off = (p >> 2) * L.XS + c + L.CW * (p & 3);
        double wv = L.PHI[off];
        for (int kp = 1; kp < pl.KS2; ++kp) wv += L.WX[(size_t)(kp - 1) * pl.NS1 * L.XS + off];
        const double mult = (double)((eh >> 16) & 255u);
        g += (mult * wv) * (xe[eh & 255u] * xe[(eh >> 8) & 255u]);
    }
    return g;
}
#endif  // BF_HOST_EMU
