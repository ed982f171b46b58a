// bfhip_oob.h -- a linear + quadratic surrogate outside its bound, without a second evaluation (gfx950; also compiled for the
// host by tests/emu).
//
// Reference (modules/poly.py:480-503, PolyModel._fj_bound): with beta = sqrt((x - mu)^T H (x - mu)) > alpha the surrogate is
// evaluated at the projected point x_0 = (alpha x + (beta - alpha) mu) / beta = mu + t (x - mu), t = alpha / beta:
//     ff = (beta ff_0 - (beta - alpha) f_mu) / alpha                                      (:487)
//     jj = jj_0 + ((ff_0 - f_mu) / alpha - jj_0 . (x - mu) / beta) * H (x - mu) / beta    (:496)
// For f(x) = c0 + lin . x + x^T S x / 2 everything at x_0 follows from S x -- which the trip has computed at x -- and S mu
// (per-dimension table, PD_SMU), because S x_0 = S mu + t (S x - S mu):
//     jj_0          = g_mu + t sv,   g_mu = S mu + lin,   sv = S x - S mu = S (x - mu)
//     ff_0          = f(mu) + t a1 + t^2 a2 / 2,   a1 = (x - mu) . g_mu,   a2 = (x - mu) . sv
//     jj_0 . (x-mu) = a1 + t a2
// a1 and a2 do not depend on beta.  The same value as the reference's to rounding (not bit for bit: it sums f at x_0 term by
// term); cubic configs are not linear in x and keep their second pass (bfhip_sampler.hip).  f(mu) is the polynomial's own
// value at mu (bf_poly_at_mu), f_mu whatever the caller's PolyModel._f_mu holds.
#pragma once
#include "bfhip_model.h"

// One division per evaluation (1 / beta; 1 / alpha comes with the model): the quotients of the reference's formulas are
// products with the reciprocals.
struct BfOob { double t, f, coef, rbeta; };

__host__ __device__ inline BfOob bf_oob_scalars(double alpha, double inv_alpha, double f_mu, double f_poly_mu, double beta, double a1, double a2) {
    BfOob o;
    o.rbeta = 1. / beta;
    o.t = alpha * o.rbeta;
    const double f0 = f_poly_mu + o.t * (a1 + 0.5 * o.t * a2);
    o.f = (beta * f0 - (beta - alpha) * f_mu) * inv_alpha;
    o.coef = (f0 - f_mu) * inv_alpha - (a1 + o.t * a2) * o.rbeta;
    return o;
}

__host__ __device__ inline double bf_oob_grad(const BfOob &o, double g_mu, double sv, double hv) {
    return (g_mu + o.t * sv) + o.coef * (hv * o.rbeta);
}
