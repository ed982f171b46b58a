// bfhip_tnuts.h -- launch arguments and helpers shared by the two tempered-NUTS kernels (bfhip_tnuts.hip: the common surrogate at
// d <= 64 with the diagonal metric; bfhip_tnuts_gen.hip: every other density and metric the NUTS kernels run on).
#pragma once
#include <cmath>
#include "bfhip_common.h"
#include "bfhip_sampler_defs.h"
#include "bfhip_wave.h"

#define TN_MAXL BFHIP_MAX_TREEDEPTH
enum { TS_LS = 0, TS_ACC, TS_E, TS_LOGP, TS_U, TS_W, TS_N };  // per-level stack scalars

struct TnutsArgs {
    bfhip_sampler_config cfg;
    int n_chain, iter_end, iter_out0, n_out, d;
    int cpg;          // chains per workgroup (8 or 4: the other waves only run matvec jobs)
    uint64_t *rng;
    double *sc, *vec, *tu, *samples, *stats, *stats_t;
    unsigned long long *n_leapfrog;
    double *scratch;  // [n_chain][4 * TN_MAXL][64] subtree stack vectors
    const double *base_S, *base_lin;  // (d,d) symmetric S_b = A_b + A_b^T, (d,)
    double base_c0, logxi;
};

__device__ inline double tn_wsum(double v) { return wave_sum(v); }   // (bfhip_wave.h: two 4 x 4 x 4 MFMAs and two row rotations)
__device__ inline double tn_logaddexp(double a, double b) {
    const double mx = a > b ? a : b, mn = a > b ? b : a;
    return (mx == -INFINITY) ? -INFINITY : mx + log1p(exp(mn - mx));
}


// bfhip_tnuts_gen.hip: the generic kernel's launcher (cubic configs, d = 128, device-side input scaling, the Gaussian link, the
// pipeline density, the full-rank metric)
int bf_tnuts_gen_launch(bfhip_ctx *ctx, const TnutsArgs &a, const double *mat);
