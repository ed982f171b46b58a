/*
 * bf_cpu_tuned.c -- a tuned CPU evaluation of the surrogate densities, for bench.py's cpu_baseline legs ONLY
 * (TEST / MEASUREMENT INFRASTRUCTURE, like the rest of oracle/).
 *
 * The parity oracle (bf_oracle.c) follows the reference statement by statement: per-config gathers and scatters,
 * separate value and Jacobian passes over the upper-triangular coefficients, strided column reads, a heap allocation
 * per temporary, an (m, d) Jacobian per evaluation of a multi-output surrogate.  That is the right checker and a weak
 * baseline (SURVEY section 8d asks for the stronger one).  This file evaluates the SAME densities
 * (core/density.py:724-754 around modules/poly.py:466-503) the way a CPU port would:
 *
 *   * single-output surrogates: every linear and quadratic config folded into ONE symmetrised dense matrix S = A + A^T
 *     and one linear vector (value and gradient from one matvec), a second matvec for the bound test, cubic-2 / cubic-3
 *     configs in their compact masked form (one pass over the j < k < l coefficients gives value and gradient);
 *   * outside the bound (modules/poly.py:480-503): linear + quadratic surrogates need no second evaluation --
 *     S x_0 follows from S x and S mu by linearity --, surrogates with cubic configs evaluate once more at x_0;
 *   * the decay term (core/density.py:740-746): one more matvec;
 *   * constraint transforms and surrogate input scaling (core/density.py:503-507, core/module.py:76-83): per-dimension loops;
 *   * a Gaussian link of the single output (core/density.py:552-560);
 *   * multi-output surrogates behind a chi-square stage and an optional prior (the pipeline density of SURVEY 8f-1): the
 *     monomial vector phi(x) once, f = C phi, w = C^T (prec r), gradient = (d phi / dx)^T w -- 4 m n_f flops instead of the
 *     reference's (m, d) Jacobian;
 *
 * no allocation per evaluation, unit-stride rows, compiled with -O3 -mavx2 -mfma.  bfo_tuned_prepare() registers a
 * density; the NUTS driver of bf_oracle.c then uses the tuned evaluation through bfo_fast_hook.  Results agree with the
 * faithful path to rounding (tests/test_oracle_golden.py: every feature set above against bfo_logp_and_grad).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bf_oracle.h"

#define TUNED_MAX 8
#define TUNED_MAXD 128

typedef struct {
    int n;              /* masked inputs */
    int order;          /* BFO_CUBIC_2 | BFO_CUBIC_3 */
    const int *mask;
    const double *a;    /* the config's coefficient block of output 0: (n,n) or (n,n,n) */
} tuned_cubic;

typedef struct {
    int i[3];           /* factors of the monomial (-1: none) */
} tuned_mono;

typedef struct {
    const bfo_density *dn;
    int d, m;
    /* single output */
    double *S;          /* (d,d) symmetric: all quadratic configs */
    double *lin;        /* (d,) all linear configs */
    double c0;
    double *Smu;        /* S mu (the bound centre) */
    double f_poly_mu;   /* c0 + lin.mu + mu.S mu / 2 */
    int n_cubic;
    tuned_cubic cubic[4];
    /* multi-output (link_kind 2) */
    int nf;             /* monomials, the constant first */
    tuned_mono *mono;
    double *C;          /* (m, nf) */
    double *Ct;         /* (nf, m) */
    double *Hs, *Hds;   /* symmetrised bound / decay Hessians */
    int decay_shared;   /* the decay term's centre and Hessian are the bound's arrays (both come from the fit points): one product serves both */
} tuned_t;

static tuned_t g_tab[TUNED_MAX];
static int g_n = 0;

extern int (*bfo_fast_hook)(const bfo_density *, const double *, int, double *, double *);

static inline double dotn(const double *a, const double *b, int n) {
    double s = 0.;
#pragma omp simd reduction(+ : s)
    for (int k = 0; k < n; ++k) s += a[k] * b[k];
    return s;
}

static inline void matvec(const double *M, const double *x, double *out, int d) {
    for (int i = 0; i < d; ++i) out[i] = dotn(M + (size_t)i * d, x, d);
}

/* cubic configs at x: value added to *f, gradient added to g (modules/_poly.pyx:49-137 in compact masked form) */
static void cubic_add(const tuned_t *t, const double *x, double *f, double *g) {
    for (int c = 0; c < t->n_cubic; ++c) {
        const tuned_cubic *cc = &t->cubic[c];
        const int n = cc->n;
        double xin[TUNED_MAXD], gin[TUNED_MAXD];
        for (int i = 0; i < n; ++i) { xin[i] = x[cc->mask[i]]; gin[i] = 0.; }
        double fs = 0.;
        if (cc->order == BFO_CUBIC_2) {
            /* f = sum_j x_j^2 sum_k a[j,k] x_k ;  df/dx_j = 2 x_j v_j + sum_k a[k,j] x_k^2 */
            double x2[TUNED_MAXD];
            for (int j = 0; j < n; ++j) x2[j] = xin[j] * xin[j];
            for (int j = 0; j < n; ++j) {
                const double *row = cc->a + (size_t)j * n;
                const double v = dotn(row, xin, n);
                fs += x2[j] * v;
                gin[j] += 2. * xin[j] * v;
                const double xj2 = x2[j];
#pragma omp simd
                for (int k = 0; k < n; ++k) gin[k] += row[k] * xj2;
            }
        } else {
            /* f = sum_{j<k<l} a[j,k,l] x_j x_k x_l : one pass, three gradient scatters */
            for (int j = 0; j < n; ++j)
                for (int k = j + 1; k < n; ++k) {
                    const double *row = cc->a + ((size_t)j * n + k) * n;
                    const double xjk = xin[j] * xin[k];
                    double s = 0.;
                    for (int l = k + 1; l < n; ++l) {
                        const double a = row[l];
                        s += a * xin[l];
                        gin[l] += a * xjk;
                    }
                    fs += s * xjk;
                    gin[j] += s * xin[k];
                    gin[k] += s * xin[j];
                }
        }
        *f += fs;
        for (int i = 0; i < n; ++i) g[cc->mask[i]] += gin[i];
    }
}

/* the single-output polynomial (all configs) at xs: value and gradient; sx_out receives S xs when not NULL */
static double poly_single(const tuned_t *t, const double *xs, double *g, double *sx_out) {
    const int d = t->d;
    double f = t->c0;
    for (int i = 0; i < d; ++i) {
        const double s = dotn(t->S + (size_t)i * d, xs, d);
        if (sx_out) sx_out[i] = s;
        g[i] = s + t->lin[i];
        f += xs[i] * (t->lin[i] + 0.5 * s);
    }
    if (t->n_cubic) cubic_add(t, xs, &f, g);
    return f;
}

/* monomials and their gradient contraction: phi (nf,) at x; grad_i = sum_p w_p d phi_p / d x_i */
static void mono_eval(const tuned_t *t, const double *x, double *phi) {
    for (int p = 0; p < t->nf; ++p) {
        const tuned_mono *mo = &t->mono[p];
        double v = 1.;
        for (int k = 0; k < 3 && mo->i[k] >= 0; ++k) v *= x[mo->i[k]];
        phi[p] = v;
    }
}

static void mono_grad(const tuned_t *t, const double *x, const double *w, double *g) {
    for (int i = 0; i < t->d; ++i) g[i] = 0.;
    for (int p = 0; p < t->nf; ++p) {
        const tuned_mono *mo = &t->mono[p];
        const double wp = w[p];
        if (mo->i[0] < 0) continue;
        if (mo->i[1] < 0) { g[mo->i[0]] += wp; continue; }
        if (mo->i[2] < 0) {
            g[mo->i[0]] += wp * x[mo->i[1]];
            g[mo->i[1]] += wp * x[mo->i[0]];
            continue;
        }
        g[mo->i[0]] += wp * (x[mo->i[1]] * x[mo->i[2]]);
        g[mo->i[1]] += wp * (x[mo->i[0]] * x[mo->i[2]]);
        g[mo->i[2]] += wp * (x[mo->i[0]] * x[mo->i[1]]);
    }
}

#define TUNED_MAXM 1024
#define TUNED_MAXNF 4096

/* the multi-output surrogate + chi-square at xs: returns like, gradient with respect to xs in g */
static double multi_eval(const tuned_t *t, const double *xs, double *g) {
    const bfo_density *dn = t->dn;
    const int m = t->m, nf = t->nf;
    double phi[TUNED_MAXNF], fm[TUNED_MAXM], r[TUNED_MAXM], w[TUNED_MAXNF];
    mono_eval(t, xs, phi);
    for (int k = 0; k < m; ++k) fm[k] = dotn(t->C + (size_t)k * nf, phi, nf);
    double q = 0.;
    for (int k = 0; k < m; ++k) fm[k] -= dn->chi2_y[k];
    if (dn->chi2_prec) for (int k = 0; k < m; ++k) r[k] = dotn(dn->chi2_prec + (size_t)k * m, fm, m);
    else for (int k = 0; k < m; ++k) r[k] = dn->chi2_prec_diag[k] * fm[k];
    q = dotn(fm, r, m);
    for (int p = 0; p < nf; ++p) w[p] = -dotn(t->Ct + (size_t)p * m, r, m);
    mono_grad(t, xs, w, g);
    return dn->link_logp0 - 0.5 * q;
}

/* the same outside the bound (modules/poly.py:480-503 per output, contracted with -r): x_0 the projected point */
static double multi_eval_oob(const tuned_t *t, const double *xs, const double *xm, const double *hv, double beta, double *g) {
    const bfo_density *dn = t->dn;
    const bfo_poly_model *pm = &dn->poly;
    const int d = t->d, m = t->m, nf = t->nf;
    const double alpha = pm->alpha;
    double x0[TUNED_MAXD], phi[TUNED_MAXNF], f0[TUNED_MAXM], fm[TUNED_MAXM], r[TUNED_MAXM], w[TUNED_MAXNF], g0[TUNED_MAXD];
    for (int i = 0; i < d; ++i) x0[i] = (alpha * xs[i] + (beta - alpha) * pm->mu[i]) / beta;
    mono_eval(t, x0, phi);
    for (int k = 0; k < m; ++k) {
        f0[k] = dotn(t->C + (size_t)k * nf, phi, nf);
        fm[k] = (beta * f0[k] - (beta - alpha) * pm->f_mu[k]) / alpha - dn->chi2_y[k];
    }
    if (dn->chi2_prec) for (int k = 0; k < m; ++k) r[k] = dotn(dn->chi2_prec + (size_t)k * m, fm, m);
    else for (int k = 0; k < m; ++k) r[k] = dn->chi2_prec_diag[k] * fm[k];
    const double q = dotn(fm, r, m);
    /* grad = -sum_o r_o (J0[o,:] + coef_o gb), coef_o = (f0_o - f_mu_o) / alpha - J0[o,:].xm / beta, gb = hv / beta */
    for (int p = 0; p < nf; ++p) w[p] = -dotn(t->Ct + (size_t)p * m, r, m);
    mono_grad(t, x0, w, g0);
    double rf = 0.;
    for (int k = 0; k < m; ++k) rf += -r[k] * (f0[k] - pm->f_mu[k]);
    const double coef = rf / alpha - dotn(g0, xm, d) / beta;
    for (int i = 0; i < d; ++i) g[i] = g0[i] + coef * (hv[i] / beta);
    return dn->link_logp0 - 0.5 * q;
}

static int tuned_eval(const bfo_density *dn, const double *x, int original_space, double *logp, double *grad) {
    const tuned_t *t = NULL;
    for (int i = 0; i < g_n; ++i)
        if (g_tab[i].dn == dn) { t = &g_tab[i]; break; }
    if (!t) return 0;
    const int d = t->d;
    const bfo_poly_model *pm = &dn->poly;
    double xo[TUNED_MAXD], jd[TUNED_MAXD], xs[TUNED_MAXD], xm[TUNED_MAXD], hv[TUNED_MAXD], g[TUNED_MAXD], sx[TUNED_MAXD];
    const int transformed = (!original_space) && dn->ranges != NULL;
    if (transformed) {
        bfo_to_original_j(x, dn->ranges, jd, dn->hard_bounds, (size_t)d);
        bfo_to_original_f(x, dn->ranges, xo, dn->hard_bounds, (size_t)d);
    } else {
        for (int i = 0; i < d; ++i) { xo[i] = x[i]; jd[i] = 1.; }
    }
    if (dn->su_lo) for (int i = 0; i < d; ++i) xs[i] = (xo[i] - dn->su_lo[i]) / dn->su_diff[i];
    else for (int i = 0; i < d; ++i) xs[i] = xo[i];
    double f;
    double beta = 0.;
    int oob = 0;
    if (pm->use_bound) {
        for (int i = 0; i < d; ++i) xm[i] = xs[i] - pm->mu[i];
        matvec(t->Hs, xm, hv, d);   /* the symmetrised Hessian: H xm = xm H */
        beta = sqrt(dotn(xm, hv, d));
        oob = beta > pm->alpha;
        if (beta != beta) return 0;    /* NaN: the faithful path decides */
    }
    if (dn->link_kind == 2) {
        f = oob ? multi_eval_oob(t, xs, xm, hv, beta, g) : multi_eval(t, xs, g);
    } else if (!oob) {
        f = poly_single(t, xs, g, NULL);
    } else {
        const double alpha = pm->alpha;
        double f0, j0[TUNED_MAXD];
        if (t->n_cubic == 0) {
            /* by linearity: x_0 = (alpha xs + (beta - alpha) mu) / beta, S x_0 = (alpha S xs + (beta - alpha) S mu) / beta */
            const double ca = alpha / beta, cb = (beta - alpha) / beta;
            double x0[TUNED_MAXD];
            matvec(t->S, xs, sx, d);
            f0 = t->c0;
            for (int i = 0; i < d; ++i) {
                x0[i] = ca * xs[i] + cb * pm->mu[i];
                const double s0 = ca * sx[i] + cb * t->Smu[i];
                j0[i] = s0 + t->lin[i];
                f0 += x0[i] * (t->lin[i] + 0.5 * s0);
            }
        } else {
            double x0[TUNED_MAXD];
            for (int i = 0; i < d; ++i) x0[i] = (alpha * xs[i] + (beta - alpha) * pm->mu[i]) / beta;
            f0 = poly_single(t, x0, j0, NULL);
        }
        f = (beta * f0 - (beta - alpha) * pm->f_mu[0]) / alpha;
        const double coef = (f0 - pm->f_mu[0]) / alpha - dotn(j0, xm, d) / beta;
        for (int i = 0; i < d; ++i) g[i] = j0[i] + coef * (hv[i] / beta);
    }
    /* core/module.py:226, density.py:558 */
    if (dn->su_diff) for (int i = 0; i < d; ++i) g[i] = g[i] / dn->su_diff[i];
    if (transformed) for (int i = 0; i < d; ++i) g[i] = g[i] * jd[i];
    if (dn->link_kind == 1) {
        const double r = f - dn->link_y, dphi = -(dn->link_prec * r);
        f = dn->link_logp0 - 0.5 * (r * (dn->link_prec * r));
        for (int i = 0; i < d; ++i) g[i] = dphi * g[i];
    }
    if (dn->link_kind == 2 && dn->prior_mu) {
        double pr = 0.;
        for (int i = 0; i < d; ++i) {
            const double dx = xo[i] - dn->prior_mu[i];
            pr += dn->prior_prec[i] * dx * dx;
            g[i] += -(dn->prior_prec[i] * dx) * jd[i];
        }
        f += dn->prior_c0 - 0.5 * pr;
    }
    if (dn->use_decay && t->decay_shared) {
        /* (x - mu_d) = xm and H_d xm = hv: the bound's product and radius (the device's two-matrix kernels do the same) */
        const double b2 = beta * beta, ex = b2 - dn->decay_alpha2;
        f -= dn->decay_gamma * (ex > 0. ? ex : (ex != ex ? ex : 0.));
        if (b2 > dn->decay_alpha2) for (int i = 0; i < d; ++i) g[i] -= 2. * dn->decay_gamma * hv[i];
    } else if (dn->use_decay) {
        double xd[TUNED_MAXD], hd[TUNED_MAXD];
        for (int i = 0; i < d; ++i) xd[i] = xo[i] - dn->decay_mu[i];
        matvec(t->Hds, xd, hd, d);
        const double b2 = dotn(xd, hd, d), ex = b2 - dn->decay_alpha2;
        f -= dn->decay_gamma * (ex > 0. ? ex : (ex != ex ? ex : 0.));
        if (b2 > dn->decay_alpha2) for (int i = 0; i < d; ++i) g[i] -= 2. * dn->decay_gamma * hd[i];
    }
    if (transformed) {
        double jj[TUNED_MAXD], s = 0.;
        bfo_to_original_jj(x, dn->ranges, jj, dn->hard_bounds, (size_t)d);
        for (int i = 0; i < d; ++i) {
            s += log(fabs(jd[i]));
            g[i] += jj[i] / jd[i];
        }
        f += s;
    }
    *logp = f;
    memcpy(grad, g, sizeof(double) * (size_t)d);
    return 1;
}

/* The bound's and the decay term's Hessians are inverses of covariance matrices (modules/poly.py:269, core/density.py:803):
 * symmetric up to the rounding of np.linalg.inv.  The tuned evaluation uses the symmetrised copy (H + H^T) / 2, so that x H and
 * H x are one product; a matrix that is not symmetric to 1e-9 of its largest entry is declined (the faithful path runs). */
static double *symmetrised(const double *M, int d) {
    double mx = 0.;
    for (int i = 0; i < d * d; ++i) mx = fabs(M[i]) > mx ? fabs(M[i]) : mx;
    for (int i = 0; i < d; ++i)
        for (int k = i + 1; k < d; ++k)
            if (!(fabs(M[(size_t)i * d + k] - M[(size_t)k * d + i]) <= 1e-9 * mx)) return NULL;
    double *S = (double *)malloc(sizeof(double) * (size_t)d * d);
    for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) S[(size_t)i * d + k] = 0.5 * (M[(size_t)i * d + k] + M[(size_t)k * d + i]);
    return S;
}

static void tuned_free(tuned_t *t) {
    free(t->S); free(t->lin); free(t->Smu); free(t->mono); free(t->C); free(t->Ct); free(t->Hs); free(t->Hds);
    memset(t, 0, sizeof(*t));
}

/* 0 on success, -1 when the density is outside the tuned forms (nothing registered: the faithful path runs) */
int bfo_tuned_prepare(const bfo_density *dn) {
    const bfo_poly_model *pm = &dn->poly;
    const int d = dn->d, m = pm->output_size;
    if (d > TUNED_MAXD || g_n >= TUNED_MAX) return -1;
    if (m != 1 && dn->link_kind != 2) return -1;
    tuned_t *t = &g_tab[g_n];
    memset(t, 0, sizeof(*t));
    if (pm->use_bound && !(t->Hs = symmetrised(pm->hess, d))) return -1;
    if (dn->use_decay && !(t->Hds = symmetrised(dn->decay_hess, d))) { tuned_free(t); return -1; }
    t->decay_shared = dn->use_decay && pm->use_bound && !dn->su_lo && pm->hess && pm->mu &&
                      memcmp(dn->decay_hess, pm->hess, (size_t)d * d * sizeof(double)) == 0 &&
                      memcmp(dn->decay_mu, pm->mu, (size_t)d * sizeof(double)) == 0;
    t->dn = dn;
    t->d = d;
    t->m = m;
    if (dn->link_kind == 2) {
        /* monomials: the constant, then every config's terms in its own packing order; identical monomials of different
         * configs stay separate columns (their coefficients add up in the product) */
        if (m > TUNED_MAXM) { tuned_free(t); return -1; }
        size_t nf = 1;
        for (int c = 0; c < pm->n_config; ++c) {
            const size_t n = (size_t)pm->configs[c].n_in;
            switch (pm->configs[c].order) {
            case BFO_LINEAR: nf += n; break;
            case BFO_QUADRATIC: nf += n * (n + 1) / 2; break;
            case BFO_CUBIC_2: nf += n * n; break;
            case BFO_CUBIC_3: nf += n * (n - 1) * (n - 2) / 6; break;
            }
        }
        if (nf > TUNED_MAXNF) { tuned_free(t); return -1; }
        t->nf = (int)nf;
        t->mono = (tuned_mono *)malloc(sizeof(tuned_mono) * nf);
        t->C = (double *)calloc((size_t)m * nf, sizeof(double));
        t->Ct = (double *)calloc((size_t)m * nf, sizeof(double));
        size_t p = 0;
        t->mono[p].i[0] = t->mono[p].i[1] = t->mono[p].i[2] = -1;
        p += 1;
        for (int c = 0; c < pm->n_config; ++c) {
            const bfo_poly_config *cf = &pm->configs[c];
            const int n = cf->n_in;
#define MONO(a, b, cidx) do { t->mono[p].i[0] = (a); t->mono[p].i[1] = (b); t->mono[p].i[2] = (cidx); } while (0)
            if (cf->order == BFO_LINEAR) {
                for (int o = 0; o < cf->n_out; ++o) t->C[(size_t)cf->out_mask[o] * nf + 0] += cf->coef[(size_t)o * (n + 1)];
                for (int j = 0; j < n; ++j, ++p) {
                    MONO(cf->in_mask[j], -1, -1);
                    for (int o = 0; o < cf->n_out; ++o) t->C[(size_t)cf->out_mask[o] * nf + p] = cf->coef[(size_t)o * (n + 1) + 1 + j];
                }
            } else if (cf->order == BFO_QUADRATIC) {
                for (int j = 0; j < n; ++j)
                    for (int k = j; k < n; ++k, ++p) {
                        MONO(cf->in_mask[j], cf->in_mask[k], -1);
                        for (int o = 0; o < cf->n_out; ++o) t->C[(size_t)cf->out_mask[o] * nf + p] = cf->coef[((size_t)o * n + j) * n + k];
                    }
            } else if (cf->order == BFO_CUBIC_2) {
                for (int j = 0; j < n; ++j)
                    for (int k = 0; k < n; ++k, ++p) {
                        MONO(cf->in_mask[j], cf->in_mask[j], cf->in_mask[k]);
                        for (int o = 0; o < cf->n_out; ++o) t->C[(size_t)cf->out_mask[o] * nf + p] = cf->coef[((size_t)o * n + j) * n + k];
                    }
            } else {
                for (int j = 0; j < n; ++j)
                    for (int k = j + 1; k < n; ++k)
                        for (int l = k + 1; l < n; ++l, ++p) {
                            MONO(cf->in_mask[j], cf->in_mask[k], cf->in_mask[l]);
                            for (int o = 0; o < cf->n_out; ++o)
                                t->C[(size_t)cf->out_mask[o] * nf + p] = cf->coef[(((size_t)o * n + j) * n + k) * n + l];
                        }
            }
#undef MONO
        }
        for (int k = 0; k < m; ++k)
            for (size_t q = 0; q < nf; ++q) t->Ct[q * m + k] = t->C[(size_t)k * nf + q];
        /* a squared factor repeats an index: mono_grad's product rule needs distinct slots, which the three-slot form gives
         * (x_j x_j x_k: slots 0 and 1 both j -> 2 x_j x_k in g_j and x_j^2 in g_k) */
    } else {
        t->S = (double *)calloc((size_t)d * d, sizeof(double));
        t->lin = (double *)calloc((size_t)d, sizeof(double));
        t->Smu = (double *)calloc((size_t)d, sizeof(double));
        for (int c = 0; c < pm->n_config; ++c) {
            const bfo_poly_config *cf = &pm->configs[c];
            const int n = cf->n_in;
            if (cf->n_out != 1 || cf->out_mask[0] != 0) { tuned_free(t); return -1; }
            if (cf->order == BFO_LINEAR) {
                t->c0 += cf->coef[0];
                for (int i = 0; i < n; ++i) t->lin[cf->in_mask[i]] += cf->coef[1 + i];
            } else if (cf->order == BFO_QUADRATIC) {
                for (int j = 0; j < n; ++j)
                    for (int k = j; k < n; ++k) { /* only j <= k is defined, modules/_poly.pyx:13-28 */
                        const double a = cf->coef[(size_t)j * n + k];
                        const int gj = cf->in_mask[j], gk = cf->in_mask[k];
                        if (j == k) t->S[(size_t)gj * d + gj] += 2. * a;
                        else { t->S[(size_t)gj * d + gk] += a; t->S[(size_t)gk * d + gj] += a; }
                    }
            } else {
                if (t->n_cubic >= 4 || n > TUNED_MAXD) { tuned_free(t); return -1; }
                tuned_cubic *cc = &t->cubic[t->n_cubic++];
                cc->n = n; cc->order = cf->order; cc->mask = cf->in_mask; cc->a = cf->coef;
            }
        }
        if (pm->use_bound) {
            matvec(t->S, pm->mu, t->Smu, d);
            t->f_poly_mu = t->c0 + dotn(t->lin, pm->mu, d) + 0.5 * dotn(pm->mu, t->Smu, d);
        }
    }
    g_n += 1;
    bfo_fast_hook = tuned_eval;
    return 0;
}

void bfo_tuned_clear(void) {
    for (int i = 0; i < g_n; ++i) tuned_free(&g_tab[i]);
    g_n = 0;
    bfo_fast_hook = NULL;
}
