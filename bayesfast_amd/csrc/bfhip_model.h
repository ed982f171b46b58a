// bfhip_model.h -- device-resident surrogate density, per-dimension table and random streams (gfx950).
// No HIP runtime dependency: tests/emu compiles the sampler's group kernel for the host with BF_HOST_EMU
// (wave emulation by fibres, test infrastructure only), where __host__ / __device__ are empty.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/bfhip.h"
#ifdef BF_HOST_EMU
#define __host__
#define __device__
#else
#include <hip/hip_runtime.h>
#endif

#ifndef BF_HOST_EMU
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
#endif

// ---- per-dimension parameter table (DP doubles each) ---------------------------------------------
enum {
    PD_KIND = 0,  // constraint kind as a double: 0 affine, 1 both bounds, 2 lower only, 3 upper only
    PD_LO,        // ranges[:,0]
    PD_RG,        // ranges[:,1] - ranges[:,0]
    PD_SU_LO,     // Surrogate.input_scales[:,0]
    PD_SU_DIFF,   // Surrogate._input_scales_diff
    PD_LIN,       // linear coefficients
    PD_MU,        // bound centre
    PD_DMU,       // decay centre
    PD_SMU,       // S mu (S = A + A^T, mu the bound centre)
    PD_HD,        // the weights of the bound proof's norm: the diagonal of the bound's Hessian (bf_bound_lam_max_weighted), or ones
    PD_HDD,       // the same for the decay term's Hessian
    PD_N
};

// ---- pipeline density (bfhip_pld.h): multi-output surrogate + Gaussian likelihood (+ prior), device-side description ----
struct PldDev {
    int on;                 // the uploaded density is a pipeline density
    int m, MP, NT1, NS2;    // rows of C' (the outputs, or the monomial count when the outputs were compressed); padded to 16; row
                            // tiles of GEMM1; k-steps of GEMM2
    int m_full;             // the surrogate's output_size
    double k_ff, k_fy;      // compressed outputs: |tail of Q^T f_mu'|^2 and (tail of Q^T f_mu') . (tail of Q^T y'), 0 otherwise
    int nf, PP, NS1, NT2;   // monomials; padded to 16; k-steps of GEMM1; row tiles of GEMM2
    int KS2, KPJ2;          // K-split of GEMM2 (partial sums in separate W slots) and k-steps per part
    int n_ent;              // entries per dimension of the gradient table
    int only8;              // the LDS block fits in the eight-chain forms' layout only (33 doubles per B-operand row): the launchers take them
    int has_prior;
    int tri;                // C' is upper triangular (the R of the output compression, bfhip_pipeline_upload): row tile t of C' is zero left of
                            // column 16 t, row tile u of C'^T right of column 16 u + 15 -- the group kernel's contractions leave those k-steps out
    const double *CF, *CTF; // A fragments (see above)
    const double *yw, *fmuw;    // (MP) whitened data vector and f_mu, zero padded
    const unsigned *mono;       // (PP) i1 | i2 << 8 | i3 << 16; index DP = 1, DP + 1 = 0 (padding monomials)
    const unsigned long long *gtab;   // [n_ent][DP]: low word = monomial p, high word = a | b << 8 | mult << 16 (padding: a = DP + 1)
    const double *prior_mu, *prior_prec;   // (DP) original-space Gaussian prior, zero padded (prec 0 = no prior on that input)
    double logp0, prior_c0;
};


// Device-resident surrogate density.  Matrices are stored as MFMA A-operand fragments:
//   frag[(t * NS + s) * 64 + l] = M[16 t + (l & 15)][4 s + (l >> 4)],  t < T = DP/16, s < NS = DP/4
// so that one wave-instruction reads 512 contiguous bytes (v_mfma_f64_16x16x4_f64, A[i=l&15][k=l>>4]).
struct DevModel {
    int d, DP;
    int has_transform, has_su, has_quad, use_bound, use_decay, has_cubic;
    int decay_shared;   // the decay term's Hessian and centre are the bound's, bit for bit (the usual case: both are taken from the fit points)
    const double *pd;   // [PD_N][DP]
    const double *Sf;   // quadratic form, symmetrised: S = A + A^T (so grad = S x + lin, f = c0 + lin.x + x.Sx/2)
    const double *Hf;   // bound Hessian
    const double *Hdf;  // decay Hessian
    double c0, alpha, f_mu, decay_alpha2, decay_gamma;
    double f_poly_mu;   // the linear + quadratic surrogate at mu (bf_poly_at_mu)
    double inv_alpha;   // 1 / alpha (0 without a bound)
    int has_link;       // the surrogate's output m feeds a Gaussian likelihood: logp = link_logp0 - link_prec (m - link_y)^2 / 2
    double link_y, link_prec, link_logp0;
    double lam_max_d;   // the same for the decay Hessian
    double lam_max;     // the bound proof's constant: (x - mu)^T H (x - mu) <= lam_max sum_j hd_j (x_j - mu_j)^2 for every x, hd = the
                        // table row PD_HD (bf_bound_lam_max_weighted); 0 without a bound
    // cubic terms in compact (masked) form; pos2/pos3 map a dimension to its index in the mask or -1
    int n2, n3;
    const int *mask2, *pos2, *mask3, *pos3;  // mask: [n], pos: [DP]
    const double *A2;    // [n2][n2] cubic-2 coefficients a[j][k]  (f = sum_j x_j^2 sum_k a[j][k] x_k)
    const double *A2t;   // [n2][n2] its transpose
    const double *T3t;   // [n3][n3][n3] symmetric fill of the j<k<l coefficients, zero where indices repeat,
                         // stored [k][l][j] so that consecutive lanes (j) read consecutive words
    PldDev pld;          // pld.on: the density is [multi-output surrogate, Gaussian likelihood, optional prior] (bfhip_pld.h);
                         // then has_quad = has_cubic = has_link = 0 and the fields above describe transforms, bound and decay
};

// ---- xoshiro256++ / splitmix64 --------------------------------------------------------------------
__host__ __device__ inline uint64_t bf_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

__host__ __device__ inline uint64_t bf_xoshiro_next(uint64_t (&s)[4]) {
    uint64_t result = bf_rotl(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = bf_rotl(s[3], 45);
    return result;
}

__host__ __device__ inline uint64_t bf_mix64(uint64_t z) {  // splitmix64 output function
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define BF_GOLDEN 0x9E3779B97F4A7C15ULL
#define BF_TWO_M53 1.1102230246251565e-16
#define BF_TWO_PI 6.283185307179586476925286766559

__host__ __device__ inline double bf_u01(uint64_t x) { return (double)(x >> 11) * BF_TWO_M53; }          // [0,1)
__host__ __device__ inline double bf_u01_open0(uint64_t x) { return (double)((x >> 11) + 1) * BF_TWO_M53; } // (0,1]

