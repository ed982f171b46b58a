// bfhip_eval.h -- surrogate-density evaluation on MFMA, "wave-local" layout.
//
// One wavefront evaluates 16 points.  Lane l = (c = l & 15, g = l >> 4) owns, for point c, the
// dimensions {4 e + g : e = 0 .. E-1}, E = DP/4.  With that ownership the B operand of
// v_mfma_f64_16x16x4_f64 for k-step s (B[k = l>>4][n = l&15] = x_c[4 s + g]) is exactly the lane's own
// element s, and the D result of row-tile t (D[row = g + 4 r][col = c]) is its own element 4 t + r:
// G^T = M . X^T maps the layout onto itself, so a matvec needs no data movement at all.
#pragma once
#include "bfhip_common.h"

// constraint transform of one coordinate: transforms/_constraint.pyx:133-215 (to_original f, j, jj)
__device__ inline void bf_to_original(double x, int kind, double lo, double rg, double &xo, double &J, double &J2) {
    double tmp, jt, j2t;
    if (kind == 1) {
        tmp = 1. / (1. + exp(-x));
        jt = tmp * (1. - tmp);
        double t2 = exp(x);
        j2t = -t2 * (t2 - 1.) / (t2 + 1.) / (t2 + 1.) / (t2 + 1.);
    } else if (kind == 2) {
        tmp = exp(x);
        jt = tmp;
        j2t = tmp;
    } else if (kind == 3) {
        double ex = exp(x);
        tmp = 1. - ex;
        jt = -ex;
        j2t = -ex;
    } else {
        tmp = x;
        jt = 1.;
        j2t = 0.;
    }
    xo = lo + tmp * rg;
    J = jt * rg;
    J2 = j2t * rg;
}

// Cubic-2 and cubic-3 contributions to gradient component `dim` of one point (modules/_poly.pyx:49-137),
// compact masked tables (DevModel).  X(k) returns x_k of the point; xj = x_dim.
//   cubic-2: f = sum_j x_j^2 v1_j, v1 = A x ;  df/dx_j = 2 x_j v1_j + (A^T x^2)_j
//   cubic-3: f = sum_{j<k<l} a x_j x_k x_l ;   df/dx_j = 1/2 sum_{k,l} T[j,k,l] x_k x_l (T symmetric fill),
//            and x . grad = 3 f (Euler), so the value needs no second pass.
template <typename XF>
__device__ inline void bf_cubic_grad(const DevModel &m, int dim, double xj, XF X, double &gc, double &fc) {
    gc = 0.;
    fc = 0.;
    if (dim >= m.DP) return;
    const int p2 = m.n2 > 0 ? m.pos2[dim] : -1;
    if (p2 >= 0) {
        double v1 = 0., v2 = 0.;
        for (int k = 0; k < m.n2; ++k) {
            const double xk = X(m.mask2[k]);
            v1 += m.A2t[k * m.n2 + p2] * xk;       // a[dim][k]
            v2 += m.A2[k * m.n2 + p2] * (xk * xk); // a[k][dim]
        }
        gc += 2. * xj * v1 + v2;
        fc += xj * xj * v1;
    }
    const int p3 = m.n3 > 0 ? m.pos3[dim] : -1;
    if (p3 >= 0) {
        double s = 0.;
        for (int k = 0; k < m.n3; ++k) {
            double t = 0.;
            for (int l = 0; l < m.n3; ++l) t += m.T3t[((size_t)k * m.n3 + l) * m.n3 + p3] * X(m.mask3[l]);
            s += t * X(m.mask3[k]);
        }
        gc += 0.5 * s;
        fc += xj * (0.5 * s) * (1. / 3.);
    }
}

// np.clip(x, 0, inf) keeps NaN
__device__ inline double bf_clip0(double x) { return x > 0. ? x : (x != x ? x : 0.); }

// out = M x for the 16 points of the wave; Af = A-fragments of M (LDS or global)
template <int T>
__device__ inline void bf_matvec_w1(const double *Af, const double (&x)[4 * T], double (&out)[4 * T], int lane) {
    constexpr int NS = 4 * T;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        d4_t acc = {0., 0., 0., 0.};
#pragma unroll
        for (int s = 0; s < NS; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[(t * NS + s) * 64 + lane], x[s], acc, 0, 0, 0);
        out[4 * t + 0] = acc[0];
        out[4 * t + 1] = acc[1];
        out[4 * t + 2] = acc[2];
        out[4 * t + 3] = acc[3];
    }
}

// sum over the four lanes (g = 0..3) that share a point
__device__ inline double bf_sum_g(double v) {
    v += bf_shfl_xor(v, 16);
    v += bf_shfl_xor(v, 32);
    return v;
}

// Density.logp_and_grad (core/density.py:724-754) for the 16 points of the wave.
// x: own elements of the input (transformed space unless original_space); padded dims hold 0.
// PL fixes the feature set of the common surrogate at compile time (linear + quadratic configs with the bound;
// no transform, input scaling, decay or cubic configs: bf_model_plain), which removes the optional features'
// register arrays -- at d = 128 the generic instantiation spills, the plain one does not.
template <int T, bool PL = false>
__device__ inline void bf_eval_w1(const DevModel &mm, const double *Sf, const double *Hf, const double *Hdf,
                                  const double *pd, int original_space, const double (&x)[4 * T], double &logp,
                                  double (&grad)[4 * T], int lane, double *xst /* wave-private LDS [16][DP], cubic only */) {
    constexpr int E = 4 * T;
    const int DP = 16 * T;
    const int g = lane >> 4;
    struct Flags {  // the model's switches, constants when PL
        bool has_transform, has_su, use_decay, has_cubic, has_quad, use_bound, has_link;
        double c0, alpha, f_mu, decay_gamma, decay_alpha2;
    };
    const Flags m = {PL ? false : (bool)mm.has_transform, PL ? false : (bool)mm.has_su, PL ? false : (bool)mm.use_decay,
                     PL ? false : (bool)mm.has_cubic, PL ? true : (bool)mm.has_quad, PL ? true : (bool)mm.use_bound,
                     PL ? false : (bool)mm.has_link, mm.c0, mm.alpha, mm.f_mu, mm.decay_gamma, mm.decay_alpha2};
    const bool tr = m.has_transform && !original_space;
    double xs[E], jac[E], gj[E], hv[E];
    double logdet = 0.;
    double bd2 = 0.;
    // ---- constraint transform + surrogate scaling (density.py:503-507, module.py:76-83) ----
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int dim = 4 * e + g;
        double xo = x[e];
        jac[e] = 1.;
        gj[e] = 0.;
        if (tr) {
            double J, J2;
            bf_to_original(x[e], (int)pd[PD_KIND * DP + dim], pd[PD_LO * DP + dim], pd[PD_RG * DP + dim], xo, J, J2);
            logdet += log(fabs(J));
            jac[e] = J;
            gj[e] = J2 / J;
        }
        hv[e] = xo;  // keep xo for the decay term
        xs[e] = m.has_su ? (xo - pd[PD_SU_LO * DP + dim]) / pd[PD_SU_DIFF * DP + dim] : xo;
    }
    // ---- decay penalty (density.py:740-746), evaluated in the ORIGINAL space ----
    double dgrad[E];
    if (m.use_decay) {
        double xd[E];
#pragma unroll
        for (int e = 0; e < E; ++e) xd[e] = hv[e] - pd[PD_DMU * DP + 4 * e + g];
        bf_matvec_w1<T>(Hdf, xd, dgrad, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) bd2 += xd[e] * dgrad[e];
    }
    // ---- polynomial (modules/poly.py:466-478) ----
    double quad = 0., lin = 0., b2 = 0.;
    double xm[E];
    if (m.has_quad) {
        bf_matvec_w1<T>(Sf, xs, grad, lane);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) grad[e] = 0.;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double c = pd[PD_LIN * DP + 4 * e + g];
        quad += xs[e] * grad[e];
        lin += c * xs[e];
        grad[e] += c;
    }
    // cubic configs: the point's coordinates go through the wave-private LDS stage
    const int pc = lane & 15;
    auto add_cubic = [&](const double (&xe)[E], double (&ge)[E]) -> double {
        double fsum = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) xst[pc * DP + 4 * e + g] = xe[e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            double gc, fc;
            bf_cubic_grad(mm, 4 * e + g, xe[e], [&](int k) { return xst[pc * DP + k]; }, gc, fc);
            ge[e] += gc;
            fsum += fc;
        }
        __builtin_amdgcn_wave_barrier();
        return bf_sum_g(fsum);
    };
    double fcub = 0.;
    if (m.has_cubic) fcub = add_cubic(xs, grad);
    if (m.use_bound) {
#pragma unroll
        for (int e = 0; e < E; ++e) xm[e] = xs[e] - pd[PD_MU * DP + 4 * e + g];
        bf_matvec_w1<T>(Hf, xm, hv, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) b2 += xm[e] * hv[e];
    }
    quad = bf_sum_g(quad);
    lin = bf_sum_g(lin);
    if (m.use_bound) b2 = bf_sum_g(b2);
    if (m.use_decay) bd2 = bf_sum_g(bd2);
    if (tr) logdet = bf_sum_g(logdet);
    double f = ((m.c0 + lin) + 0.5 * quad) + fcub;
    // ---- linear extrapolation outside the alpha-ellipsoid (modules/poly.py:480-503) ----
    if (m.use_bound) {
        const double beta = sqrt(b2);
        const bool oob = beta > m.alpha;
        if (__any(oob)) {  // wave-uniform: the second matvec runs for the whole tile, rarely
            double x0[E], j0[E];
#pragma unroll
            for (int e = 0; e < E; ++e)
                x0[e] = oob ? (m.alpha * xs[e] + (beta - m.alpha) * pd[PD_MU * DP + 4 * e + g]) / beta : xs[e];
            if (m.has_quad) {
                bf_matvec_w1<T>(Sf, x0, j0, lane);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) j0[e] = 0.;
            }
            double quad0 = 0., lin0 = 0., dotj = 0., fcub0 = 0.;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double c = pd[PD_LIN * DP + 4 * e + g];
                quad0 += x0[e] * j0[e];
                lin0 += c * x0[e];
                j0[e] += c;
            }
            if (m.has_cubic) fcub0 = add_cubic(x0, j0);
#pragma unroll
            for (int e = 0; e < E; ++e) dotj += j0[e] * xm[e];
            quad0 = bf_sum_g(quad0);
            lin0 = bf_sum_g(lin0);
            dotj = bf_sum_g(dotj);
            if (oob) {
                const double f0 = ((m.c0 + lin0) + 0.5 * quad0) + fcub0;
                f = (beta * f0 - (beta - m.alpha) * m.f_mu) / m.alpha;
                const double coef = (f0 - m.f_mu) / m.alpha - dotj / beta;
#pragma unroll
                for (int e = 0; e < E; ++e) grad[e] = j0[e] + coef * (hv[e] / beta);
            }
        }
    }
    // ---- chain rule (module.py:226, density.py:558), decay, transform Jacobian (density.py:747-750) ----
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (m.has_su) grad[e] = grad[e] / pd[PD_SU_DIFF * DP + 4 * e + g];
        grad[e] = grad[e] * jac[e];
    }
    if (m.has_link) {  // the next module of the pipeline (density.py:552-560): logp = phi(m), grad = phi'(m) grad m
        const double r = f - mm.link_y, dphi = -(mm.link_prec * r);
        f = mm.link_logp0 - 0.5 * (r * (mm.link_prec * r));
#pragma unroll
        for (int e = 0; e < E; ++e) grad[e] = dphi * grad[e];
    }
    if (m.use_decay) {
        f -= m.decay_gamma * bf_clip0(bd2 - m.decay_alpha2);
        if (bd2 > m.decay_alpha2) {
#pragma unroll
            for (int e = 0; e < E; ++e) grad[e] -= 2. * m.decay_gamma * dgrad[e];
        }
    }
    if (tr) {
        f += logdet;
#pragma unroll
        for (int e = 0; e < E; ++e) grad[e] += gj[e];
    }
    logp = f;
}
