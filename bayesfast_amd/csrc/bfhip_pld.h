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
// whitening), which removes the m x m product from the kernel: what is left are two dense contractions on the FP64 matrix
// cores with the 16 chains as the 16 columns,
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

// LDS regions of the pipeline block (doubles), in this order behind `base`
struct PldLds {
    double *XE;    // [16][DP + 2]  evaluation point of every chain, then 1 and 0
    double *CH;    // [16]          beta of the chains outside the bound's ellipsoid, 0 inside
    double *YW;    // [2][MP]       y' and f_mu'
    double *RED;   // [NT1][2][16]  per row tile and chain: sum r^2, sum (f_0 - f_mu) r
    double *RB;    // [NS2][XS]     r as the B operand of GEMM2
    double *PHI;   // [NS1][XS]     monomials as the B operand of GEMM1; slot 0 of W afterwards
    double *WX;    // [KS2 - 1][NS1][XS]  further partial-sum slots of W
};

__host__ __device__ inline size_t pld_lds_doubles(int DP, int MP, int PP, int KS2) {
    const size_t ns1 = PP / 4, ns2 = MP / 4, nt1 = MP / 16;
    return (size_t)16 * (DP + 2) + 16 + (size_t)2 * MP + nt1 * 32 + ns2 * PLD_XS + (size_t)KS2 * ns1 * PLD_XS;
}

#ifndef BF_HOST_EMU
__device__ inline PldLds pld_lds(double *base, int DP, const PldDev &pl) {
    PldLds L;
    L.XE = base;
    L.CH = L.XE + 16 * (DP + 2);
    L.YW = L.CH + 16;
    L.RED = L.YW + 2 * pl.MP;
    L.RB = L.RED + pl.NT1 * 32;
    L.PHI = L.RB + (size_t)pl.NS2 * PLD_XS;
    L.WX = L.PHI + (size_t)pl.NS1 * PLD_XS;
    return L;
}

// once per launch, all threads: y' and f_mu' into LDS
__device__ inline void pld_stage(const PldDev &pl, const PldLds &L, int tid, int nth) {
    for (int i = tid; i < pl.MP; i += nth) {
        L.YW[i] = pl.yw[i];
        L.YW[pl.MP + i] = pl.fmuw[i];
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
            const unsigned mo = pl.mono[p];
            const double v = (xe[mo & 255u] * xe[(mo >> 8) & 255u]) * xe[(mo >> 16) & 255u];
            L.PHI[(p >> 2) * PLD_XS + c + 16 * (p & 3)] = v;
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

// GEMM1 and its epilogue, all NWV waves of the workgroup: F_0 = C' Phi per row tile, the bound's extrapolation per chain,
// r = F - y' into the B operand of GEMM2, and the tile's contributions to sum r^2 and sum (f_0 - f_mu) r
__device__ inline void pld_gemm1(const PldDev &pl, const PldLds &L, double alpha, int w, int nwv, int lane) {
    const int mc = lane & 15, mg = lane >> 4;
    const double beta = L.CH[mc];
    for (int t = w; t < pl.NT1; t += nwv) {
        const d4_t acc = pld_tile(pl.CF + (size_t)t * pl.NS1 * 64, L.PHI, pl.NS1, lane);
        double s_rr = 0., s_fr = 0.;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int row = 16 * t + 4 * r4 + mg;
            const double f0 = acc[r4], y = L.YW[row], fmu = L.YW[pl.MP + row];
            const double fv = beta > 0. ? (beta * f0 - (beta - alpha) * fmu) / alpha : f0;   // modules/poly.py:487
            const double r = fv - y;
            L.RB[(4 * t + r4) * PLD_XS + lane] = r;   // row >> 2 = 4 t + r4, 16 (row & 3) + chain = lane
            s_rr += r * r;
            s_fr += (f0 - fmu) * r;
        }
        s_rr = pld_rowsum4(s_rr);
        s_fr = pld_rowsum4(s_fr);
        if (lane < 16) {
            L.RED[(t * 2 + 0) * 16 + lane] = s_rr;
            L.RED[(t * 2 + 1) * 16 + lane] = s_fr;
        }
    }
}

// GEMM2, all waves: W = C'^T R, (row tile, K part) jobs; part kp lands in W slot kp (slot 0 = PHI, which GEMM1 has consumed)
__device__ inline void pld_gemm2(const PldDev &pl, const PldLds &L, int w, int nwv, int lane) {
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

// chain wave c after GEMM2: the chain's sums (lane t holds row tile t's parts; the caller reduces over the wave) ...
__device__ inline void pld_sums(const PldDev &pl, const PldLds &L, int c, int lane, double &s_rr, double &s_fr) {
    s_rr = s_fr = 0.;
    for (int t = lane; t < pl.NT1; t += 64) {
        s_rr += L.RED[(t * 2 + 0) * 16 + c];
        s_fr += L.RED[(t * 2 + 1) * 16 + c];
    }
}

// ... and component `dim` of J_0^T r: the monomials that contain x_dim, each times its cofactor
__device__ inline double pld_grad(const PldDev &pl, const PldLds &L, int DP, int c, int dim) {
    const double *xe = L.XE + c * (DP + 2);
    double g = 0.;
    for (int i = 0; i < pl.n_ent; ++i) {
        const unsigned long long en = pl.gtab[(size_t)i * DP + dim];
        const unsigned eh = (unsigned)(en >> 32);
        const int p = (int)(unsigned)en, off = (p >> 2) * PLD_XS + c + 16 * (p & 3);
        double wv = L.PHI[off];
        for (int kp = 1; kp < pl.KS2; ++kp) wv += L.WX[(size_t)(kp - 1) * pl.NS1 * PLD_XS + off];
        const double mult = (double)((eh >> 16) & 255u);
        g += (mult * wv) * (xe[eh & 255u] * xe[(eh >> 8) & 255u]);
    }
    return g;
}
#endif  // BF_HOST_EMU
