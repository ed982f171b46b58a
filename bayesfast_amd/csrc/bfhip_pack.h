// bfhip_pack.h -- host-side re-layout of a bfhip_density_desc into the device model's tables (no HIP dependency;
// shared by bfhip_density_upload and the host emulation of the group kernel under tests/emu).
#pragma once
#include <cmath>
#include <cstdlib>
#include <vector>
#include "bfhip_model.h"
#include "bfhip_tune.h"

static inline int padded_tiles(int d) { return d <= 16 ? 1 : d <= 32 ? 2 : d <= 64 ? 4 : 8; }

// M (d,d) row-major (optionally transposed) -> A-fragments of the DP x DP zero-padded matrix
static inline void to_fragments(const double *M, int d, int DP, bool transpose, double *frag) {
    const int T = DP / 16, NS = DP / 4;
    for (int t = 0; t < T; ++t)
        for (int s = 0; s < NS; ++s)
            for (int l = 0; l < 64; ++l) {
                int row = 16 * t + (l & 15), col = 4 * s + (l >> 4);
                double v = 0.;
                if (row < d && col < d) v = transpose ? M[(size_t)col * d + row] : M[(size_t)row * d + col];
                frag[((size_t)t * NS + s) * 64 + l] = v;
            }
}


// pd table [PD_N][DP] followed by the A-operand fragments of S, H and H_decay^T ([DP*DP] each); returns DP
static inline int bf_pack_density(const bfhip_density_desc *ds, std::vector<double> &h) {
    const int d = ds->d;
    const int T = padded_tiles(d), DP = 16 * T;
    const size_t MAT = (size_t)DP * DP;
    const size_t n_dbl = (size_t)PD_N * DP + 3 * MAT;
    h.assign(n_dbl, 0.);
    double *pd = h.data(), *Sf = pd + (size_t)PD_N * DP, *Hf = Sf + MAT, *Hdf = Hf + MAT;
    for (int i = 0; i < DP; ++i) {
        pd[PD_RG * DP + i] = 1.;
        pd[PD_SU_DIFF * DP + i] = 1.;
        pd[PD_HD * DP + i] = 1.;
        pd[PD_HDD * DP + i] = 1.;
    }
    // (tuning switch: BFHIP_NO_PROOF_WEIGHTS=1 keeps the plain norm in both proofs)
    const bool no_weights = bf_tune().no_proof_weights != 0;   // (tuning switch: the plain norm in both proofs)
    if (ds->use_bound) {   // the weights of the bound proof's norm (bf_bound_lam_max_weighted)
        bool ok = !no_weights;
        for (int i = 0; i < d; ++i) ok = ok && ds->hess[(size_t)i * d + i] > 0. && std::isfinite(ds->hess[(size_t)i * d + i]);
        if (ok)
            for (int i = 0; i < d; ++i) pd[PD_HD * DP + i] = ds->hess[(size_t)i * d + i];
    }
    if (ds->use_decay) {
        bool ok = !no_weights;
        for (int i = 0; i < d; ++i) ok = ok && ds->decay_hess[(size_t)i * d + i] > 0. && std::isfinite(ds->decay_hess[(size_t)i * d + i]);
        if (ok)
            for (int i = 0; i < d; ++i) pd[PD_HDD * DP + i] = ds->decay_hess[(size_t)i * d + i];
    }
    for (int i = 0; i < d; ++i) {
        if (ds->ranges) {
            int lo = ds->hard_bounds ? ds->hard_bounds[2 * i] : 0, hi = ds->hard_bounds ? ds->hard_bounds[2 * i + 1] : 0;
            pd[PD_KIND * DP + i] = (lo && hi) ? 1. : (lo ? 2. : (hi ? 3. : 0.));
            pd[PD_LO * DP + i] = ds->ranges[2 * i];
            pd[PD_RG * DP + i] = ds->ranges[2 * i + 1] - ds->ranges[2 * i];
        }
        if (ds->su_lo) {
            pd[PD_SU_LO * DP + i] = ds->su_lo[i];
            pd[PD_SU_DIFF * DP + i] = ds->su_diff[i];
        }
        if (ds->lin) pd[PD_LIN * DP + i] = ds->lin[i];
        if (ds->use_bound) pd[PD_MU * DP + i] = ds->mu[i];
        if (ds->use_decay) pd[PD_DMU * DP + i] = ds->decay_mu[i];
    }
    if (ds->quad) {
        // S = A + A^T from the upper triangle the reference reads (modules/_poly.pyx:13-43)
        std::vector<double> S((size_t)d * d, 0.);
        for (int j = 0; j < d; ++j)
            for (int k = j; k < d; ++k) {
                double a = ds->quad[(size_t)j * d + k];
                if (j == k) S[(size_t)j * d + j] = 2. * a;
                else { S[(size_t)j * d + k] = a; S[(size_t)k * d + j] = a; }
            }
        to_fragments(S.data(), d, DP, false, Sf);
        // S mu: outside the bound the sampler kernels get S x_0 of the projected point x_0 = mu + (alpha / beta) (x - mu)
        // from S x by linearity instead of a second pass over the tiles (bfhip_nuts_pipe.h, bfhip_group.h)
        if (ds->use_bound)
            for (int j = 0; j < d; ++j) {
                double s = 0.;
                for (int k = 0; k < d; ++k) s += S[(size_t)j * d + k] * ds->mu[k];
                pd[PD_SMU * DP + j] = s;
            }
    }
    if (ds->use_bound) to_fragments(ds->hess, d, DP, false, Hf);
    // decay gradient is (x - mu) H, i.e. H^T (x - mu): core/density.py:745
    if (ds->use_decay) to_fragments(ds->decay_hess, d, DP, true, Hdf);

    return DP;
}

// The linear + quadratic surrogate at the bound's centre, c0 + lin . mu + mu^T A mu (upper triangle of A, modules/_poly.pyx:13-43):
// the start of the expansion of f(x_0) along the ray.  NOT the descriptor's f_mu, which is whatever the caller's
// PolyModel._f_mu holds and enters the reference's formulas as such (modules/poly.py:487,496).
static inline double bf_poly_at_mu(const bfhip_density_desc *ds) {
    if (!ds->use_bound) return 0.;
    const int d = ds->d;
    double f = ds->c0;
    for (int j = 0; j < d; ++j) {
        if (ds->lin) f += ds->lin[j] * ds->mu[j];
        if (ds->quad)
            for (int k = j; k < d; ++k) f += ds->quad[(size_t)j * d + k] * ds->mu[j] * ds->mu[k];
    }
    return f;
}

// An upper bound c of the largest eigenvalue of the symmetric part of H, PROVEN by a successful Cholesky factorisation of
// c' I - sym(H) for a c' slightly below c (so that (x - mu)^T H (x - mu) <= c |x - mu|^2 for every x): a power iteration
// proposes, the factorisation disposes; the Gershgorin bound is the fallback.  Host side, once per upload.
static inline double bf_bound_lam_max(const double *hess, int d) {
    std::vector<double> A((size_t)d * d), v(d), w(d), L((size_t)d * d);
    // (a start vector without symmetry: the all-ones vector lies in the null space of every H with zero row sums)
    for (int i = 0; i < d; ++i) v[i] = 1. + 0.37 * std::sin(1.7 * (double)i + 0.3);
    double gersh = 0.;
    bool finite = true;
    for (int i = 0; i < d; ++i) {
        double rs = 0.;
        for (int k = 0; k < d; ++k) {
            A[(size_t)i * d + k] = 0.5 * (hess[(size_t)i * d + k] + hess[(size_t)k * d + i]);
            rs += std::fabs(A[(size_t)i * d + k]);
        }
        finite = finite && std::isfinite(rs);
        if (rs > gersh) gersh = rs;
    }
    if (!finite) return NAN;  // (no bound: `lam_max * r2 < alpha^2` is false for every r2 and the proof never claims anything)
    if (!(gersh > 0.)) return 0.;
    double lam = 0.;
    for (int it = 0; it < 200; ++it) {
        double nn = 0.;
        for (int i = 0; i < d; ++i) {
            double s = 0.;
            for (int k = 0; k < d; ++k) s += A[(size_t)i * d + k] * v[k];
            w[i] = s;
            nn += s * s;
        }
        nn = std::sqrt(nn);
        if (!(nn > 0.)) break;
        lam = nn;
        for (int i = 0; i < d; ++i) v[i] = w[i] / nn;
    }
    // the search starts above zero whatever the iteration found (lam = 0: v hit the null space) and ends after a bounded
    // number of steps: 1.05^n covers 1e-6 gersh .. gersh in 284 steps
    double c0 = lam * 1.01;
    if (!(c0 > 1e-6 * gersh)) c0 = 1e-6 * gersh;
    int n_try = 0;
    for (double c = c0; c < gersh && n_try < 400; c *= 1.05, ++n_try) {
        bool ok = true;
        for (int i = 0; i < d && ok; ++i)
            for (int k = 0; k <= i; ++k) {
                double s = (i == k ? c : 0.) - A[(size_t)i * d + k];
                for (int q = 0; q < k; ++q) s -= L[(size_t)i * d + q] * L[(size_t)k * d + q];
                if (i == k) {
                    if (!(s > 1e-12 * c)) { ok = false; break; }
                    L[(size_t)i * d + i] = std::sqrt(s);
                } else {
                    L[(size_t)i * d + k] = s / L[(size_t)k * d + k];
                }
            }
        if (ok) return c * (1. + 1e-9);
    }
    return gersh * (1. + 1e-9);
}

// The bound proof with a weighted norm: (x - mu)^T H (x - mu) = y^T N y <= lam_max(N) |y|^2 with y_j = sqrt(hd_j) (x_j - mu_j) and
// N = diag(hd)^-1/2 sym(H) diag(hd)^-1/2, hd > 0 (bf_pack_density takes the diagonal of H, or ones).  For a Hessian that is a
// well-conditioned matrix seen through per-dimension scales -- the bound of a surrogate whose input scales were folded in, any bound
// fitted in parameters of different units -- lam_max(N) |y|^2 is as tight as the plain norm is for the unscaled matrix, where
// lam_max(H) |x - mu|^2 overshoots by the square of the scales' spread.  Returns the proven constant for the weights hd (d of them).
static inline double bf_bound_lam_max_weighted(const double *hess, int d, const double *hd) {
    std::vector<double> N((size_t)d * d);
    for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) N[(size_t)i * d + k] = hess[(size_t)i * d + k] / std::sqrt(hd[i] * hd[k]);
    return bf_bound_lam_max(N.data(), d);
}

// stream = global chain index, so results do not depend on how chains are sharded over GPUs
static inline void bf_seed_state(uint64_t seed, uint64_t stream, uint64_t s[4]) {
    uint64_t x = seed ^ (0xD1B54A32D192ED03ULL * (stream + 1));
    for (int i = 0; i < 4; ++i) {
        x += BF_GOLDEN;
        s[i] = bf_mix64(x);
    }
}
