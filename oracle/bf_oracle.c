/*
 * bf_oracle.c -- CPU restatement of the BayesFast hot path (TEST INFRASTRUCTURE ONLY; see bf_oracle.h).
 *
 * One chain at a time, recursive NUTS tree exactly as the reference builds it, so that a replay of the
 * reference's logged random draws reproduces its trajectories.  Citations are reference file:line.
 */
#include "bf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ================= polynomial kernels: bayesfast/modules/_poly.pyx ================================= */

/* _poly.pyx:13-28 */
void bfo_quadratic_f(const double *x, const double *a, double *out, int m, int n) {
    for (int i = 0; i < m; ++i) {
        const double *ai = a + (size_t)i * n * n;
        double o = 0.;
        for (int j = 0; j < n; ++j) {
            double t = 0.;
            for (int k = j; k < n; ++k) t += ai[j * n + k] * x[k];
            o += t * x[j];
        }
        out[i] = o;
    }
}

/* _poly.pyx:34-43 */
void bfo_quadratic_j(const double *x, const double *a, double *out, int m, int n) {
    for (int i = 0; i < m; ++i) {
        const double *ai = a + (size_t)i * n * n;
        for (int j = 0; j < n; ++j) {
            double o = 2 * ai[j * n + j] * x[j];
            for (int k = 0; k < j; ++k) o += ai[k * n + j] * x[k];
            for (int k = j + 1; k < n; ++k) o += ai[j * n + k] * x[k];
            out[i * n + j] = o;
        }
    }
}

/* _poly.pyx:49-65 */
void bfo_cubic_2_f(const double *x, const double *a, double *out, int m, int n) {
    for (int i = 0; i < m; ++i) {
        const double *ai = a + (size_t)i * n * n;
        double o = 0.;
        for (int j = 0; j < n; ++j) {
            double t = 0.;
            for (int k = 0; k < n; ++k) t += ai[j * n + k] * x[k];
            o += t * x[j] * x[j];
        }
        out[i] = o;
    }
}

/* _poly.pyx:71-80 */
void bfo_cubic_2_j(const double *x, const double *a, double *out, int m, int n) {
    for (int i = 0; i < m; ++i) {
        const double *ai = a + (size_t)i * n * n;
        for (int j = 0; j < n; ++j) {
            double o = 0.;
            for (int k = 0; k < n; ++k) o += ai[j * n + k] * x[k];
            o *= 2. * x[j];
            for (int k = 0; k < n; ++k) o += ai[k * n + j] * x[k] * x[k];
            out[i * n + j] = o;
        }
    }
}

/* _poly.pyx:86-108 */
void bfo_cubic_3_f(const double *x, const double *a, double *out, int m, int n) {
    for (int i = 0; i < m; ++i) {
        const double *ai = a + (size_t)i * n * n * n;
        double o = 0.;
        for (int j = 0; j + 2 < n; ++j) {
            double s = 0.;
            for (int k = j + 1; k + 1 < n; ++k) {
                double t = 0.;
                for (int l = k + 1; l < n; ++l) t += ai[((size_t)j * n + k) * n + l] * x[l];
                s += t * x[k];
            }
            o += s * x[j];
        }
        out[i] = o;
    }
}

/* _poly.pyx:114-137 */
void bfo_cubic_3_j(const double *x, const double *a, double *out, int m, int n) {
    for (int i = 0; i < m; ++i) {
        const double *ai = a + (size_t)i * n * n * n;
        for (int j = 0; j < n; ++j) {
            double o = 0.;
            for (int k = 0; k < j; ++k) {
                double t = 0.;
                for (int l = k + 1; l < j; ++l) t += ai[((size_t)k * n + l) * n + j] * x[l];
                o += t * x[k];
                t = 0.;
                for (int l = j + 1; l < n; ++l) t += ai[((size_t)k * n + j) * n + l] * x[l];
                o += t * x[k];
            }
            for (int k = j + 1; k < n; ++k) {
                double t = 0.;
                for (int l = k + 1; l < n; ++l) t += ai[((size_t)j * n + k) * n + l] * x[l];
                o += t * x[k];
            }
            out[i * n + j] = o;
        }
    }
}

/* _poly.pyx:143-151 ; x (m,n) -> out (m, n(n+1)/2) */
void bfo_lsq_quadratic(const double *x, double *out, int m, int n) {
    size_t w = (size_t)n * (n + 1) / 2;
    for (int i = 0; i < m; ++i) {
        size_t j = 0;
        for (int k = 0; k < n; ++k)
            for (int l = k; l < n; ++l) out[i * w + j++] = x[(size_t)i * n + k] * x[(size_t)i * n + l];
    }
}

/* _poly.pyx:157-164 */
void bfo_lsq_cubic_2(const double *x, double *out, int m, int n) {
    size_t w = (size_t)n * n;
    for (int i = 0; i < m; ++i) {
        size_t j = 0;
        for (int k = 0; k < n; ++k)
            for (int l = 0; l < n; ++l)
                out[i * w + j++] = x[(size_t)i * n + k] * x[(size_t)i * n + k] * x[(size_t)i * n + l];
    }
}

/* _poly.pyx:170-177 */
void bfo_lsq_cubic_3(const double *x, double *out, int m, int n) {
    size_t w = (size_t)n * (n - 1) * (n - 2) / 6;
    for (int i = 0; i < m; ++i) {
        size_t j = 0;
        for (int k = 0; k < n; ++k)
            for (int l = k + 1; l < n; ++l)
                for (int p = l + 1; p < n; ++p)
                    out[i * w + j++] = x[(size_t)i * n + k] * x[(size_t)i * n + l] * x[(size_t)i * n + p];
    }
}

/* _poly.pyx:183-189 (coef is NOT zeroed here; PolyConfig._set allocates with np.empty and only the
 * written entries are ever read by the kernels above) */
void bfo_set_quadratic(const double *a, double *coef, int n) {
    size_t i = 0;
    for (int j = 0; j < n; ++j)
        for (int k = j; k < n; ++k) coef[j * n + k] = a[i++];
}

/* _poly.pyx:195-201 */
void bfo_set_cubic_2(const double *a, double *coef, int n) {
    size_t i = 0;
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k) coef[j * n + k] = a[i++];
}

/* _poly.pyx:207-214 */
void bfo_set_cubic_3(const double *a, double *coef, int n) {
    size_t i = 0;
    for (int j = 0; j < n; ++j)
        for (int k = j + 1; k < n; ++k)
            for (int l = k + 1; l < n; ++l) coef[((size_t)j * n + k) * n + l] = a[i++];
}

/* ================= constraint transforms: bayesfast/transforms/_constraint.pyx ===================== */

#define HB(i, s) (hb ? hb[2 * (i) + (s)] : 0)

/* _constraint.pyx:19-38 */
int bfo_from_original_f(const double *x, const double *ranges, double *out, const uint8_t *hb, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        double tmp = (x[i] - ranges[2 * i]) / (ranges[2 * i + 1] - ranges[2 * i]);
        if (HB(i, 0) && HB(i, 1)) {
            if (tmp <= 0. || tmp >= 1.) return (int)i + 1;
            tmp = log(tmp / (1. - tmp));
        } else if (HB(i, 0) && !HB(i, 1)) {
            if (tmp <= 0.) return (int)i + 1;
            tmp = log(tmp);
        } else if (!HB(i, 0) && HB(i, 1)) {
            if (tmp >= 1.) return (int)i + 1;
            tmp = log(1. - tmp);
        }
        out[i] = tmp;
    }
    return 0;
}

/* _constraint.pyx:55-77 */
int bfo_from_original_j(const double *x, const double *ranges, double *out, const uint8_t *hb, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        double tmp = (x[i] - ranges[2 * i]) / (ranges[2 * i + 1] - ranges[2 * i]);
        if (HB(i, 0) && HB(i, 1)) {
            if (tmp <= 0. || tmp >= 1.) return (int)i + 1;
            tmp = 1. / tmp / (1. - tmp);
        } else if (HB(i, 0) && !HB(i, 1)) {
            if (tmp <= 0.) return (int)i + 1;
            tmp = 1. / tmp;
        } else if (!HB(i, 0) && HB(i, 1)) {
            if (tmp >= 1.) return (int)i + 1;
            tmp = 1. / (tmp - 1.);
        } else {
            tmp = 1.;
        }
        tmp /= (ranges[2 * i + 1] - ranges[2 * i]);
        out[i] = tmp;
    }
    return 0;
}

/* _constraint.pyx:94-116 */
int bfo_from_original_jj(const double *x, const double *ranges, double *out, const uint8_t *hb, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        double tmp = (x[i] - ranges[2 * i]) / (ranges[2 * i + 1] - ranges[2 * i]);
        if (HB(i, 0) && HB(i, 1)) {
            if (tmp <= 0. || tmp >= 1.) return (int)i + 1;
            tmp = (2. * tmp - 1.) / tmp / tmp / (1. - tmp) / (1. - tmp);
        } else if (HB(i, 0) && !HB(i, 1)) {
            if (tmp <= 0.) return (int)i + 1;
            tmp = -1. / tmp / tmp;
        } else if (!HB(i, 0) && HB(i, 1)) {
            if (tmp >= 1.) return (int)i + 1;
            tmp = 1. / (tmp - 1.) / (1. - tmp);
        } else {
            tmp = 0.;
        }
        tmp /= (ranges[2 * i + 1] - ranges[2 * i]) * (ranges[2 * i + 1] - ranges[2 * i]);
        out[i] = tmp;
    }
    return 0;
}

/* _constraint.pyx:133-147 */
void bfo_to_original_f(const double *x, const double *ranges, double *out, const uint8_t *hb, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        double tmp = x[i];
        if (HB(i, 0) && HB(i, 1)) tmp = 1. / (1. + exp(-tmp));
        else if (HB(i, 0) && !HB(i, 1)) tmp = exp(tmp);
        else if (!HB(i, 0) && HB(i, 1)) tmp = 1. - exp(tmp);
        out[i] = ranges[2 * i] + tmp * (ranges[2 * i + 1] - ranges[2 * i]);
    }
}

/* _constraint.pyx:164-181 */
void bfo_to_original_j(const double *x, const double *ranges, double *out, const uint8_t *hb, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        double tmp = x[i];
        if (HB(i, 0) && HB(i, 1)) {
            tmp = 1. / (1. + exp(-tmp));
            tmp = tmp * (1. - tmp);
        } else if (HB(i, 0) && !HB(i, 1)) tmp = exp(tmp);
        else if (!HB(i, 0) && HB(i, 1)) tmp = -exp(tmp);
        else tmp = 1.;
        out[i] = tmp * (ranges[2 * i + 1] - ranges[2 * i]);
    }
}

/* _constraint.pyx:198-215 */
void bfo_to_original_jj(const double *x, const double *ranges, double *out, const uint8_t *hb, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        double tmp = x[i];
        if (HB(i, 0) && HB(i, 1)) {
            double t2 = exp(tmp);
            tmp = -t2 * (t2 - 1.) / (t2 + 1.) / (t2 + 1.) / (t2 + 1.);
        } else if (HB(i, 0) && !HB(i, 1)) tmp = exp(tmp);
        else if (!HB(i, 0) && HB(i, 1)) tmp = -exp(tmp);
        else tmp = 0.;
        out[i] = tmp * (ranges[2 * i + 1] - ranges[2 * i]);
    }
}

/* ================= PolyModel evaluation: bayesfast/modules/poly.py ================================= */

/* poly.py:430-441 (_eval_one) + 340-428 (per-order dispatch); f (n_out,), j (n_out,n_in) */
static void poly_eval_one(const bfo_poly_config *cf, const double *x, double *f, double *j) {
    int n = cf->n_in, m = cf->n_out;
    double *xin = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) xin[i] = x[cf->in_mask[i]]; /* poly.py:431 */
    switch (cf->order) {
    case BFO_LINEAR: /* poly.py:340-352: coef (m, n+1), column 0 is the constant */
        for (int i = 0; i < m; ++i) {
            const double *c = cf->coef + (size_t)i * (n + 1);
            if (f) {
                double s = 0.;
                for (int k = 0; k < n; ++k) s += c[1 + k] * xin[k];
                f[i] = s + c[0];
            }
            if (j) for (int k = 0; k < n; ++k) j[i * n + k] = c[1 + k];
        }
        break;
    case BFO_QUADRATIC:
        if (f) bfo_quadratic_f(xin, cf->coef, f, m, n);
        if (j) bfo_quadratic_j(xin, cf->coef, j, m, n);
        break;
    case BFO_CUBIC_2:
        if (f) bfo_cubic_2_f(xin, cf->coef, f, m, n);
        if (j) bfo_cubic_2_j(xin, cf->coef, j, m, n);
        break;
    case BFO_CUBIC_3:
        if (f) bfo_cubic_3_f(xin, cf->coef, f, m, n);
        if (j) bfo_cubic_3_j(xin, cf->coef, j, m, n);
        break;
    }
    free(xin);
}

/* poly.py:470-477: sum over configs with scatter-add through the masks */
static void poly_eval_all(const bfo_poly_model *pm, const double *x, double *f, double *j) {
    int d = pm->input_size, m = pm->output_size;
    if (f) memset(f, 0, sizeof(double) * (size_t)m);
    if (j) memset(j, 0, sizeof(double) * (size_t)m * d);
    for (int c = 0; c < pm->n_config; ++c) {
        const bfo_poly_config *cf = &pm->configs[c];
        double *ff = f ? (double *)malloc(sizeof(double) * (size_t)cf->n_out) : NULL;
        double *jj = j ? (double *)malloc(sizeof(double) * (size_t)cf->n_out * cf->n_in) : NULL;
        poly_eval_one(cf, x, ff, jj);
        for (int o = 0; o < cf->n_out; ++o) {
            if (f) f[cf->out_mask[o]] += ff[o];
            if (j)
                for (int k = 0; k < cf->n_in; ++k)
                    j[(size_t)cf->out_mask[o] * d + cf->in_mask[k]] += jj[o * cf->n_in + k];
        }
        free(ff);
        free(jj);
    }
}

/* (x-mu)^T H (x-mu) in the reference's evaluation order: np.dot(np.dot(x - mu, hess), x - mu), poly.py:468 */
static double mahalanobis2(const double *x, const double *mu, const double *hess, int d, double *hv /* (d,) out: (x-mu) H */) {
    double s = 0.;
    for (int k = 0; k < d; ++k) {
        double t = 0.;
        for (int i = 0; i < d; ++i) t += (x[i] - mu[i]) * hess[(size_t)i * d + k];
        if (hv) hv[k] = t;
        s += t * (x[k] - mu[k]);
    }
    return s;
}

/* poly.py:466-503 */
void bfo_poly_fun_and_jac(const bfo_poly_model *pm, const double *x, double *f, double *j) {
    int d = pm->input_size, m = pm->output_size;
    if (pm->use_bound) {
        double beta = sqrt(mahalanobis2(x, pm->mu, pm->hess, d, NULL));
        if (beta > pm->alpha) { /* poly.py:480-503 (_fj_bound) */
            double alpha = pm->alpha;
            double *x0 = (double *)malloc(sizeof(double) * (size_t)d);
            double *f0 = (double *)malloc(sizeof(double) * (size_t)m);
            double *j0 = j ? (double *)malloc(sizeof(double) * (size_t)m * d) : NULL;
            double *gb = (double *)malloc(sizeof(double) * (size_t)d);
            for (int i = 0; i < d; ++i) x0[i] = (alpha * x[i] + (beta - alpha) * pm->mu[i]) / beta; /* :482 */
            poly_eval_all(pm, x0, f0, j0);
            if (f)
                for (int o = 0; o < m; ++o) f[o] = (beta * f0[o] - (beta - alpha) * pm->f_mu[o]) / alpha; /* :487 */
            if (j) {
                /* grad_beta = np.dot(hess, x - mu) / beta, poly.py:490 */
                for (int i = 0; i < d; ++i) {
                    double t = 0.;
                    for (int k = 0; k < d; ++k) t += pm->hess[(size_t)i * d + k] * (x[k] - pm->mu[k]);
                    gb[i] = t / beta;
                }
                for (int o = 0; o < m; ++o) { /* :495-496 */
                    double dot = 0.;
                    for (int k = 0; k < d; ++k) dot += j0[(size_t)o * d + k] * (x[k] - pm->mu[k]);
                    double coef = (f0[o] - pm->f_mu[o]) / alpha - dot / beta;
                    for (int k = 0; k < d; ++k) j[(size_t)o * d + k] = j0[(size_t)o * d + k] + coef * gb[k];
                }
            }
            free(x0); free(f0); free(j0); free(gb);
            return;
        }
    }
    poly_eval_all(pm, x, f, j);
}

/* ================= Density.logp_and_grad: bayesfast/core/density.py:724-754 ======================= */

/* measurement hook (bf_cpu_tuned.c): a registered tuned evaluation of the same density; NULL unless bench.py's CPU baseline
 * asked for it.  Returns 0 when it declines a point (outside the bound), and the statement-by-statement path below runs. */
int (*bfo_fast_hook)(const bfo_density *, const double *, int, double *, double *) = NULL;

void bfo_logp_and_grad(const bfo_density *dn, const double *x, int original_space, double *logp, double *grad) {
    if (bfo_fast_hook && bfo_fast_hook(dn, x, original_space, logp, grad)) return;
    int d = dn->d;
    double *xo = (double *)malloc(sizeof(double) * (size_t)d * 5);
    double *jd = xo + d, *xs = xo + 2 * d, *g = xo + 3 * d, *hv = xo + 4 * d;
    int transformed = (!original_space) && dn->ranges != NULL;
    /* density.py:503-507: j = diag(to_original_grad(x)) ; x = to_original(x) */
    if (transformed) {
        bfo_to_original_j(x, dn->ranges, jd, dn->hard_bounds, (size_t)d);
        bfo_to_original_f(x, dn->ranges, xo, dn->hard_bounds, (size_t)d);
    } else {
        for (int i = 0; i < d; ++i) { xo[i] = x[i]; jd[i] = 1.; }
    }
    /* core/module.py:76-83: surrogate input scaling */
    for (int i = 0; i < d; ++i) xs[i] = dn->su_lo ? (xo[i] - dn->su_lo[i]) / dn->su_diff[i] : xo[i];
    double f;
    if (dn->link_kind == 2) {
        /* [surrogate (m outputs), Gaussian likelihood, optional prior module]: density.py:527-560 */
        int m = dn->poly.output_size;
        double *fm = (double *)malloc(sizeof(double) * ((size_t)m * 2 + (size_t)m * d));
        double *r = fm + m, *J = fm + 2 * m;
        bfo_poly_fun_and_jac(&dn->poly, xs, fm, J);
        /* module.py:226 on every row of the surrogate's Jacobian, then density.py:558 with the diagonal input Jacobian */
        for (int k = 0; k < m; ++k)
            for (int i = 0; i < d; ++i) {
                double v = J[(size_t)k * d + i];
                if (dn->su_diff) v = v / dn->su_diff[i];
                J[(size_t)k * d + i] = v * jd[i];
            }
        double q = 0.;
        for (int k = 0; k < m; ++k) {
            double acc = 0.;
            if (dn->chi2_prec) for (int l = 0; l < m; ++l) acc += dn->chi2_prec[(size_t)k * m + l] * (fm[l] - dn->chi2_y[l]);
            else acc = dn->chi2_prec_diag[k] * (fm[k] - dn->chi2_y[k]);
            r[k] = acc;
        }
        for (int k = 0; k < m; ++k) q += (fm[k] - dn->chi2_y[k]) * r[k];
        f = dn->link_logp0 - 0.5 * q;
        for (int i = 0; i < d; ++i) {
            double acc = 0.;
            for (int k = 0; k < m; ++k) acc += -r[k] * J[(size_t)k * d + i];   /* dot(J_out (1,m), J_in (m,d)) */
            g[i] = acc;
        }
        if (dn->prior_mu) {
            double pr = 0.;
            for (int i = 0; i < d; ++i) {
                double dx = xo[i] - dn->prior_mu[i];
                pr += dn->prior_prec[i] * dx * dx;
                g[i] += -(dn->prior_prec[i] * dx) * jd[i];   /* the 'x' rows of the module's input Jacobian are diag(jd) */
            }
            f += dn->prior_c0 - 0.5 * pr;
        }
        free(fm);
    } else {
    bfo_poly_fun_and_jac(&dn->poly, xs, &f, g);
    /* core/module.py:226 (jac / input_scales_diff) then density.py:558 (chain rule with diag j) */
    for (int i = 0; i < d; ++i) {
        if (dn->su_diff) g[i] = g[i] / dn->su_diff[i];
        g[i] = g[i] * jd[i];
    }
    if (dn->link_kind == 1) { /* the next module of the pipeline, density.py:552-560: var_dict._jac[n] = dot(J_out, J_in) */
        double r = f - dn->link_y, dphi = -(dn->link_prec * r);
        f = dn->link_logp0 - 0.5 * (r * (dn->link_prec * r));
        for (int i = 0; i < d; ++i) g[i] = dphi * g[i];
    }
    }
    if (dn->use_decay) { /* density.py:740-746 */
        double beta2 = mahalanobis2(xo, dn->decay_mu, dn->decay_hess, d, hv);
        double ex = beta2 - dn->decay_alpha2;
        f -= dn->decay_gamma * (ex > 0. ? ex : (ex != ex ? ex : 0.)); /* np.clip keeps NaN */
        if (beta2 > dn->decay_alpha2)
            for (int i = 0; i < d; ++i) g[i] -= 2 * dn->decay_gamma * hv[i];
    }
    if (!original_space) { /* density.py:747-750 */
        if (dn->ranges) {
            double s = 0.;
            bfo_to_original_jj(x, dn->ranges, hv, dn->hard_bounds, (size_t)d);
            for (int i = 0; i < d; ++i) {
                s += log(fabs(jd[i]));
                g[i] += hv[i] / jd[i];
            }
            f += s;
        }
        /* identity transform: log|1| = 0 and 0/1 = 0, density.py:93-111 */
    }
    *logp = f;
    for (int i = 0; i < d; ++i) grad[i] = g[i];
    free(xo);
}

/* ================= random streams ================================================================== */

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* xoshiro256++ 1.0 (Blackman & Vigna, public domain algorithm); the device kernels use the same recurrence */
uint64_t bfo_xoshiro_next(uint64_t s[4]) {
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* stream = global chain index: results do not depend on how chains are sharded over GPUs */
void bfo_xoshiro_seed(uint64_t seed, uint64_t stream, uint64_t s[4]) {
    uint64_t x = seed ^ (0xD1B54A32D192ED03ULL * (stream + 1));
    for (int i = 0; i < 4; ++i) s[i] = splitmix64(&x);
}

#define TWO_M53 1.1102230246251565e-16 /* 2^-53 */
#define TWO_PI 6.283185307179586476925286766559

double bfo_rng_uniform(bfo_rng *r) {
    if (r->kind == 1) {
        if (r->i_uniform >= r->n_uniforms) { r->exhausted = 1; return 0.5; }
        return r->uniforms[r->i_uniform++];
    }
    return (double)(bfo_xoshiro_next(r->s) >> 11) * TWO_M53; /* [0,1) */
}

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* n standard normals.  xoshiro mode (the device kernels' definition): ONE xoshiro draw K keys a SplitMix64
 * counter stream u_i = mix64(K + (i+1) GOLDEN); pair P = (u_2P, u_2P+1) gives elements 2P (cos) and 2P+1
 * (sin) by Box-Muller, so any lane can produce its own dimensions without walking the stream. */
void bfo_rng_normal(bfo_rng *r, double *out, int n) {
    if (r->kind == 1) {
        for (int i = 0; i < n; ++i) {
            if (r->i_normal >= r->n_normals) { r->exhausted = 1; out[i] = 0.; continue; }
            out[i] = r->normals[r->i_normal++];
        }
        return;
    }
    const uint64_t K = bfo_xoshiro_next(r->s);
    for (int i = 0; i < n; i += 2) {
        const uint64_t P = (uint64_t)(i / 2);
        double u1 = (double)((mix64(K + (2 * P + 1) * 0x9E3779B97F4A7C15ULL) >> 11) + 1) * TWO_M53; /* (0,1] */
        double u2 = (double)(mix64(K + (2 * P + 2) * 0x9E3779B97F4A7C15ULL) >> 11) * TWO_M53;       /* [0,1) */
        double rad = sqrt(-2. * log(u1));
        double th = TWO_PI * u2;
        out[i] = rad * cos(th);
        if (i + 1 < n) out[i + 1] = rad * sin(th);
    }
}

/* ================= chain state ===================================================================== */

bfo_chain *bfo_chain_new(int d, const double *x0, double step_size, int adapt_step, double target,
                         double gamma, double k, double t_0, const double *metric_var, int adapt_metric,
                         const double *initial_mean, double initial_weight, long adapt_window,
                         long update_window, int doubling) {
    bfo_chain *c = (bfo_chain *)calloc(1, sizeof(bfo_chain));
    c->d = d;
    /* samplers/sample_trace.py:365-373 + step_size.py:12-23 */
    double initial_step = step_size / pow((double)d, 0.25);
    c->log_step = log(initial_step);
    c->log_bar = c->log_step;
    c->hbar = 0.;
    c->mu = log(10. * initial_step);
    c->target = target; c->gamma = gamma; c->k = k; c->t_0 = t_0;
    c->count = 1;
    c->adapt_step = adapt_step;
    /* samplers/sample_trace.py:424-455 + metrics.py:148-181 */
    double *buf = (double *)calloc((size_t)d * 8, sizeof(double));
    c->var = buf; c->std = buf + d; c->inv_std = buf + 2 * d;
    c->fg_mean = buf + 3 * d; c->fg_raw = buf + 4 * d; c->bg_mean = buf + 5 * d; c->bg_raw = buf + 6 * d;
    c->q = buf + 7 * d;
    for (int i = 0; i < d; ++i) {
        c->var[i] = metric_var ? metric_var[i] : 1.;
        c->std[i] = sqrt(c->var[i]);
        c->inv_std[i] = 1. / c->std[i];
        c->q[i] = x0[i];
        /* _WeightedVariance(n, initial_mean, initial_var, initial_weight): metrics.py:335-352 */
        c->fg_mean[i] = initial_mean ? initial_mean[i] : x0[i];
        c->fg_raw[i] = c->var[i] * initial_weight;
        /* background = _WeightedVariance(n): default initial_weight = 10, zero mean and raw_var */
        c->bg_mean[i] = 0.; c->bg_raw[i] = 0.;
    }
    c->adapt_metric = adapt_metric;
    c->initial_weight = initial_weight;
    c->fg_n = initial_weight;
    c->bg_n = 10.;
    c->n_samples = 0; c->previous_update = 0;
    c->adapt_window = adapt_window; c->update_window = update_window; c->doubling = doubling;
    c->i_iter = 0;
    return c;
}

void bfo_chain_free(bfo_chain *c) {
    if (!c) return;
    free(c->var);
    free(c->cov);
    free(c);
}

/* scipy.linalg.cholesky(a, lower=True) (LAPACK dpotrf): reads the lower triangle of a, writes l (zeros above the
 * diagonal).  Returns -1 if a pivot is not positive (LinAlgError). */
static int chol_lower(const double *a, double *l, int d) {
    for (int i = 0; i < d * d; ++i) l[i] = 0.;
    for (int j = 0; j < d; ++j) {
        double s = a[j * d + j];
        for (int k = 0; k < j; ++k) s -= l[j * d + k] * l[j * d + k];
        if (!(s > 0.)) return -1;
        double ljj = sqrt(s);
        l[j * d + j] = ljj;
        for (int i = j + 1; i < d; ++i) {
            double t = a[i * d + j];
            for (int k = 0; k < j; ++k) t -= l[i * d + k] * l[j * d + k];
            l[i * d + j] = t / ljj;
        }
    }
    return 0;
}

int bfo_chain_set_full(bfo_chain *c, const double *cov0) {
    int d = c->d;
    size_t n = (size_t)d * d;
    double *buf = (double *)calloc(n * 5, sizeof(double));
    c->cov = buf; c->chol = buf + n; c->fg_cov = buf + 2 * n; c->bg_cov = buf + 3 * n; c->work = buf + 4 * n;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) c->cov[i * d + j] = cov0 ? cov0[i * d + j] : (i == j ? 1. : 0.);
    if (chol_lower(c->cov, c->chol, d)) return -1;
    /* _WeightedCovariance(n, initial_mean, initial_cov, initial_weight): raw_cov = cov * weight (metrics.py:382-395);
     * background = _WeightedCovariance(n): weight 10, zero mean, raw_cov = 10 I */
    for (size_t i = 0; i < n; ++i) c->fg_cov[i] = c->cov[i] * c->initial_weight;
    for (int i = 0; i < d; ++i) c->bg_cov[i * d + i] = 10.;
    c->full = 1;
    return 0;
}

/* metric.velocity: metrics.py:66-71 (diag), :113-115 (full, np.dot(cov, x)) */
static void metric_velocity(const bfo_chain *c, const double *p, double *v) {
    int d = c->d;
    if (!c->full) {
        for (int i = 0; i < d; ++i) v[i] = c->var[i] * p[i];
        return;
    }
    for (int i = 0; i < d; ++i) {
        double s = 0.;
        for (int k = 0; k < d; ++k) s += c->cov[i * d + k] * p[k];
        v[i] = s;
    }
}

/* metric.random: metrics.py:83-86 (diag), :123-127 (full: solve_triangular(chol.T, normals)) */
static void metric_random(const bfo_chain *c, bfo_rng *rng, double *p) {
    int d = c->d;
    bfo_rng_normal(rng, p, d);
    if (!c->full) {
        for (int i = 0; i < d; ++i) p[i] = c->inv_std[i] * p[i];
        return;
    }
    for (int j = d - 1; j >= 0; --j) {  /* L^T p = z: the column sweep of BLAS dtrsv (upper, no transpose) */
        p[j] = p[j] / c->chol[j * d + j];
        for (int i = 0; i < j; ++i) p[i] -= c->chol[j * d + i] * p[j];
    }
}

/* metrics.py:354-360 */
static void welford_add(double *mean, double *raw, double *n, const double *x, int d) {
    *n += 1.;
    for (int i = 0; i < d; ++i) {
        double old_diff = x[i] - mean[i];
        mean[i] += old_diff / *n;
        double new_diff = x[i] - mean[i];
        raw[i] += 1. * old_diff * new_diff;
    }
}

/* _WeightedCovariance.add_sample: metrics.py:401-407 */
static void welford_cov_add(double *mean, double *raw, double *n, const double *x, int d, double *old_diff) {
    *n += 1.;
    for (int i = 0; i < d; ++i) {
        old_diff[i] = x[i] - mean[i];
        mean[i] += old_diff[i] / *n;
    }
    for (int i = 0; i < d; ++i) {
        double new_diff = x[i] - mean[i];
        for (int j = 0; j < d; ++j) raw[i * d + j] += 1. * new_diff * old_diff[j];
    }
}

/* QuadMetricFullAdapt.update: metrics.py:294-324 */
static void metric_update_full(bfo_chain *c, const double *sample) {
    int d = c->d;
    size_t n = (size_t)d * d;
    long delta = c->n_samples - c->previous_update;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)d);
    welford_cov_add(c->fg_mean, c->fg_cov, &c->fg_n, sample, d, tmp);
    welford_cov_add(c->bg_mean, c->bg_cov, &c->bg_n, sample, d, tmp);
    free(tmp);
    if ((delta + 1) % c->update_window == 0) { /* _update_from_weightvar: :287-292 */
        for (size_t i = 0; i < n; ++i) c->cov[i] = c->fg_cov[i] / c->fg_n;
        if (chol_lower(c->cov, c->work, d) == 0) memcpy(c->chol, c->work, sizeof(double) * n);
    }
    if (delta >= c->adapt_window) {
        memcpy(c->fg_mean, c->bg_mean, sizeof(double) * (size_t)d);
        memcpy(c->fg_cov, c->bg_cov, sizeof(double) * n);
        c->fg_n = c->bg_n;
        for (int i = 0; i < d; ++i) c->bg_mean[i] = 0.;
        for (size_t i = 0; i < n; ++i) c->bg_cov[i] = 0.;
        for (int i = 0; i < d; ++i) c->bg_cov[i * d + i] = 10.;
        c->bg_n = 10.;
        c->previous_update = c->n_samples;
        if (c->doubling) c->adapt_window *= 2;
    }
    c->n_samples += 1;
}

/* metrics.py:186-211 */
static void metric_update(bfo_chain *c, const double *sample, int warmup) {
    if (!warmup || !c->adapt_metric) return;
    if (c->full) { metric_update_full(c, sample); return; }
    int d = c->d;
    long delta = c->n_samples - c->previous_update;
    welford_add(c->fg_mean, c->fg_raw, &c->fg_n, sample, d);
    welford_add(c->bg_mean, c->bg_raw, &c->bg_n, sample, d);
    if ((delta + 1) % c->update_window == 0) { /* metrics.py:181-184 */
        for (int i = 0; i < d; ++i) {
            c->var[i] = c->fg_raw[i] / c->fg_n;
            c->std[i] = sqrt(c->var[i]);
            c->inv_std[i] = 1. / c->std[i];
        }
    }
    if (delta >= c->adapt_window) {
        memcpy(c->fg_mean, c->bg_mean, sizeof(double) * (size_t)d);
        memcpy(c->fg_raw, c->bg_raw, sizeof(double) * (size_t)d);
        c->fg_n = c->bg_n;
        for (int i = 0; i < d; ++i) { c->bg_mean[i] = 0.; c->bg_raw[i] = 0.; }
        c->bg_n = 10.;
        c->previous_update = c->n_samples;
        if (c->doubling) c->adapt_window *= 2;
    }
    c->n_samples += 1;
}

/* step_size.py:31-45 */
static void step_size_update(bfo_chain *c, double accept_stat, int warmup) {
    if (!warmup) return;
    if (!c->adapt_step) return;
    double count = (double)c->count;
    double w = 1. / (count + c->t_0);
    c->hbar = ((1. - w) * c->hbar + w * (c->target - accept_stat));
    c->log_step = c->mu - c->hbar * sqrt(count) / c->gamma;
    double mk = pow(count, -c->k);
    c->log_bar = mk * c->log_step + (1. - mk) * c->log_bar;
    c->count += 1;
}

/* ================= leapfrog: samplers/hmc_utils/integration.py:28-34,68-95 ======================== */

typedef struct {
    double *q, *p, *v, *grad; /* (d,) each, one allocation at q */
    double energy, logp;
    double u, vt, weight;     /* tempered samplers (TState, integration.py:98-222): tempering coordinate, its momentum, P(beta=1|x) */
} lf_state;

static lf_state state_alloc(int d) {
    lf_state s;
    s.q = (double *)malloc(sizeof(double) * (size_t)d * 4);
    s.p = s.q + d; s.v = s.q + 2 * d; s.grad = s.q + 3 * d;
    s.energy = s.logp = 0.;
    s.u = s.vt = s.weight = 0.;
    return s;
}
static void state_free(lf_state *s) { free(s->q); s->q = NULL; }
static lf_state state_clone(const lf_state *a, int d) {
    lf_state s = state_alloc(d);
    memcpy(s.q, a->q, sizeof(double) * (size_t)d * 4);
    s.energy = a->energy; s.logp = a->logp;
    s.u = a->u; s.vt = a->vt; s.weight = a->weight;
    return s;
}

void bfo_leapfrog(const bfo_density *dn, const double *var, double eps, const double *q, const double *p,
                  const double *grad, double *q_new, double *p_new, double *v_new, double *grad_new,
                  double *energy_new, double *logp_new) {
    int d = dn->d;
    double dt = 0.5 * eps;
    for (int i = 0; i < d; ++i) p_new[i] = p[i] + dt * grad[i];         /* integration.py:80 */
    for (int i = 0; i < d; ++i) v_new[i] = var[i] * p_new[i];           /* :82 */
    for (int i = 0; i < d; ++i) q_new[i] = q[i] + eps * v_new[i];       /* :85 */
    bfo_logp_and_grad(dn, q_new, 0, logp_new, grad_new);                /* :87 */
    for (int i = 0; i < d; ++i) p_new[i] = p_new[i] + dt * grad_new[i]; /* :90 */
    double kin = 0.;
    for (int i = 0; i < d; ++i) { v_new[i] = var[i] * p_new[i]; kin += p_new[i] * v_new[i]; } /* :92, metrics.py:88-91 */
    *energy_new = 0.5 * kin - *logp_new;                                 /* :93 */
}

void bfo_leapfrog_full(const bfo_density *dn, const double *cov, double eps, const double *q, const double *p,
                       const double *grad, double *q_new, double *p_new, double *v_new, double *grad_new,
                       double *energy_new, double *logp_new) {
    bfo_chain c;
    memset(&c, 0, sizeof(c));
    c.d = dn->d; c.full = 1; c.cov = (double *)cov;
    int d = dn->d;
    double dt = 0.5 * eps;
    for (int i = 0; i < d; ++i) p_new[i] = p[i] + dt * grad[i];
    metric_velocity(&c, p_new, v_new);
    for (int i = 0; i < d; ++i) q_new[i] = q[i] + eps * v_new[i];
    bfo_logp_and_grad(dn, q_new, 0, logp_new, grad_new);
    for (int i = 0; i < d; ++i) p_new[i] = p_new[i] + dt * grad_new[i];
    metric_velocity(&c, p_new, v_new);
    double kin = 0.;
    for (int i = 0; i < d; ++i) kin += p_new[i] * v_new[i];
    *energy_new = 0.5 * kin - *logp_new;
}

/* the integrator step with the chain's own metric */
static void leapfrog_chain(const bfo_density *dn, const bfo_chain *c, double eps, const double *q, const double *p,
                           const double *grad, double *q_new, double *p_new, double *v_new, double *grad_new,
                           double *energy_new, double *logp_new) {
    if (c->full) bfo_leapfrog_full(dn, c->cov, eps, q, p, grad, q_new, p_new, v_new, grad_new, energy_new, logp_new);
    else bfo_leapfrog(dn, c->var, eps, q, p, grad, q_new, p_new, v_new, grad_new, energy_new, logp_new);
}

/* ================= NUTS tree: samplers/nuts.py ==================================================== */

typedef struct {
    int has;            /* 0 for the divergence stub Subtree(None, ...) of nuts.py:130 */
    lf_state left, right;
    double *p_sum;      /* (d,) */
    double *prop_q;     /* (d,) proposal position */
    double prop_energy, prop_logp;
    double prop_u, prop_weight; /* TProposal, tnuts.py:12-19 */
    double log_size, accept_sum;
    long n_proposals;
} subtree;

typedef struct {
    const bfo_density *dn;
    bfo_chain *ch;
    bfo_rng *rng;
    int d;
    double max_change, start_energy, max_energy_change;
    long n_leapfrog;
    int err; /* -3 logbern(NaN) */
    /* tempered samplers: the base density and log xi (base_hmc.py:220-231); NULL for NUTS / HMC */
    const bfo_density *base;
    double logxi;
} tree_ctx;

static void tempered_step(const bfo_density *dn, const bfo_density *base, double logxi, const bfo_chain *c, double eps,
                          const lf_state *s0, lf_state *s1);

static void subtree_free(subtree *t) {
    if (t->has) {
        state_free(&t->left);
        state_free(&t->right);
        free(t->p_sum);
        free(t->prop_q);
    }
    t->has = 0;
}

/* nuts.py:200-203 */
static int logbern(tree_ctx *cx, double l) {
    if (isnan(l)) { cx->err = -3; return 0; }
    return log(bfo_rng_uniform(cx->rng)) < l;
}

static double dot(const double *a, const double *b, int d) {
    double s = 0.;
    for (int i = 0; i < d; ++i) s += a[i] * b[i];
    return s;
}

/* optional diagnostics: energies of every leaf, in integration order */
double *bfo_trace_buf = NULL;
long bfo_trace_cap = 0, bfo_trace_n = 0;
void bfo_set_trace(double *buf, long cap) { bfo_trace_buf = buf; bfo_trace_cap = cap; bfo_trace_n = 0; }

/* nuts.py:105-132 */
static subtree single_step(tree_ctx *cx, const lf_state *left, double eps, int *diverging) {
    int d = cx->d;
    subtree t;
    memset(&t, 0, sizeof(t));
    lf_state right = state_alloc(d);
    if (cx->base) tempered_step(cx->dn, cx->base, cx->logxi, cx->ch, eps, left, &right);
    else leapfrog_chain(cx->dn, cx->ch, eps, left->q, left->p, left->grad, right.q, right.p, right.v, right.grad,
                        &right.energy, &right.logp);
    cx->n_leapfrog += 1;
    if (bfo_trace_buf && bfo_trace_n + 8 <= bfo_trace_cap) {
        double *t = bfo_trace_buf + bfo_trace_n;
        bfo_trace_n += 8;
        t[0] = 0.; t[1] = right.energy; t[2] = eps; t[3] = t[4] = t[5] = t[6] = t[7] = 0.;
    }
    double energy_change = right.energy - cx->start_energy;
    if (isnan(energy_change)) energy_change = INFINITY;
    if (fabs(energy_change) > fabs(cx->max_energy_change)) cx->max_energy_change = energy_change;
    if (fabs(energy_change) < cx->max_change) {
        double p_accept = exp(-energy_change);
        if (p_accept > 1.) p_accept = 1.;
        t.has = 1;
        t.left = right;
        t.right = state_clone(&right, d);
        t.p_sum = (double *)malloc(sizeof(double) * (size_t)d);
        memcpy(t.p_sum, right.p, sizeof(double) * (size_t)d);
        t.prop_q = (double *)malloc(sizeof(double) * (size_t)d);
        memcpy(t.prop_q, right.q, sizeof(double) * (size_t)d);
        t.prop_energy = right.energy; t.prop_logp = right.logp;
        t.prop_u = right.u; t.prop_weight = right.weight;
        t.log_size = -energy_change; t.accept_sum = p_accept; t.n_proposals = 1;
        *diverging = 0;
        return t;
    }
    state_free(&right);
    t.has = 0; t.log_size = -INFINITY; t.accept_sum = 0.; t.n_proposals = 1;
    *diverging = 1;
    return t;
}

/* nuts.py:134-178 */
static subtree build_subtree(tree_ctx *cx, const lf_state *left, int depth, double eps, int *diverging, int *turning) {
    int d = cx->d;
    if (depth == 0) {
        *turning = 0;
        return single_step(cx, left, eps, diverging);
    }
    subtree t1 = build_subtree(cx, left, depth - 1, eps, diverging, turning);
    if (*diverging || *turning) return t1;
    subtree t2 = build_subtree(cx, &t1.right, depth - 1, eps, diverging, turning);
    subtree t;
    memset(&t, 0, sizeof(t));
    t.has = 1;
    t.left = state_clone(&t1.left, d);
    /* right = tree2.right, which is None for a divergence stub; it is never read in that case */
    t.right = t2.has ? state_clone(&t2.right, d) : state_clone(&t1.right, d);
    t.p_sum = (double *)malloc(sizeof(double) * (size_t)d);
    t.prop_q = (double *)malloc(sizeof(double) * (size_t)d);
    if (!(*diverging || *turning)) {
        for (int i = 0; i < d; ++i) t.p_sum[i] = t1.p_sum[i] + t2.p_sum[i];
        int turn = (dot(t.p_sum, t.left.v, d) <= 0) || (dot(t.p_sum, t.right.v, d) <= 0);
        if (bfo_trace_buf && bfo_trace_n + 8 <= bfo_trace_cap) {
            double *tt = bfo_trace_buf + bfo_trace_n;
            bfo_trace_n += 8;
            tt[0] = 1.; tt[1] = depth; tt[2] = dot(t.p_sum, t.left.v, d); tt[3] = dot(t.p_sum, t.right.v, d);
            tt[4] = tt[5] = tt[6] = tt[7] = 0.;
        }
        if (depth > 1) { /* nuts.py:154-161 */
            double *ps = (double *)malloc(sizeof(double) * (size_t)d);
            for (int i = 0; i < d; ++i) ps[i] = t1.p_sum[i] + t2.left.p[i];
            int turn1 = (dot(ps, t1.left.v, d) <= 0) || (dot(ps, t2.left.v, d) <= 0);
            for (int i = 0; i < d; ++i) ps[i] = t1.right.p[i] + t2.p_sum[i];
            int turn2 = (dot(ps, t1.right.v, d) <= 0) || (dot(ps, t2.right.v, d) <= 0);
            turn = turn | turn1 | turn2;
            free(ps);
        }
        *turning = turn;
        /* np.logaddexp(tree1.log_size, tree2.log_size), nuts.py:163 */
        double a = t1.log_size, b = t2.log_size, mx = a > b ? a : b, mn = a > b ? b : a;
        t.log_size = (mx == -INFINITY) ? -INFINITY : mx + log1p(exp(mn - mx));
        if (logbern(cx, t2.log_size - t.log_size)) { /* nuts.py:164-167 */
            memcpy(t.prop_q, t2.prop_q, sizeof(double) * (size_t)d);
            t.prop_energy = t2.prop_energy; t.prop_logp = t2.prop_logp;
            t.prop_u = t2.prop_u; t.prop_weight = t2.prop_weight;
        } else {
            memcpy(t.prop_q, t1.prop_q, sizeof(double) * (size_t)d);
            t.prop_energy = t1.prop_energy; t.prop_logp = t1.prop_logp;
            t.prop_u = t1.prop_u; t.prop_weight = t1.prop_weight;
        }
    } else { /* nuts.py:168-171 */
        memcpy(t.p_sum, t1.p_sum, sizeof(double) * (size_t)d);
        t.log_size = t1.log_size;
        memcpy(t.prop_q, t1.prop_q, sizeof(double) * (size_t)d);
        t.prop_energy = t1.prop_energy; t.prop_logp = t1.prop_logp;
        t.prop_u = t1.prop_u; t.prop_weight = t1.prop_weight;
    }
    t.accept_sum = t1.accept_sum + t2.accept_sum;
    t.n_proposals = t1.n_proposals + t2.n_proposals;
    subtree_free(&t1);
    subtree_free(&t2);
    return t;
}

/* BaseHMC.astep with NUTS._hamiltonian_step; base_hmc.py:62-85, nuts.py:205-217, 24-103 */
static void tempered_state(const bfo_density *dn, const bfo_density *base, double logxi, const bfo_chain *c, lf_state *s);

static int nuts_iteration_t(const bfo_density *dn, bfo_chain *c, bfo_rng *rng, int warmup, int max_treedepth,
                            double max_change, double *sample_out, double *st, long *n_leapfrog, const bfo_density *base,
                            double logxi, double *u_io, double *st_t) {
    int d = c->d;
    tree_ctx cx;
    cx.dn = dn; cx.ch = c; cx.rng = rng; cx.d = d; cx.max_change = max_change; cx.max_energy_change = 0.;
    cx.n_leapfrog = 0; cx.err = 0;
    cx.base = base; cx.logxi = logxi;
    /* p0 = metric.random(rng): metrics.py:83-86 */
    lf_state start = state_alloc(d);
    metric_random(c, rng, start.p);
    memcpy(start.q, c->q, sizeof(double) * (size_t)d);
    if (base) { /* BaseTHMC.astep: base_hmc.py:233-262: u continues from the last iteration, v0 is one more normal draw */
        start.u = *u_io;
        bfo_rng_normal(rng, &start.vt, 1);
        tempered_state(dn, base, logxi, c, &start);
    } else {
    /* integrator.compute_state: integration.py:28-34 */
    bfo_logp_and_grad(dn, start.q, 0, &start.logp, start.grad);
    double kin = 0.;
    metric_velocity(c, start.p, start.v);
    for (int i = 0; i < d; ++i) kin += start.p[i] * start.v[i];
    start.energy = 0.5 * kin - start.logp;
    }
    if (!isfinite(start.energy)) { state_free(&start); return -1; } /* base_hmc.py:72-76 */
    double step_size = exp(warmup ? c->log_step : c->log_bar);      /* step_size.py:25-29 */

    /* Tree.__init__: nuts.py:24-43 */
    cx.start_energy = start.energy;
    lf_state left = state_clone(&start, d), right = state_clone(&start, d);
    double *prop_q = (double *)malloc(sizeof(double) * (size_t)d * 3);
    double *p_sum = prop_q + d, *tmp = prop_q + 2 * d;
    memcpy(prop_q, start.q, sizeof(double) * (size_t)d);
    memcpy(p_sum, start.p, sizeof(double) * (size_t)d);
    double prop_energy = start.energy, prop_logp = start.logp;
    double prop_u = start.u, prop_weight = start.weight;
    int depth = 0;
    double log_size = 0., accept_sum = 0.;
    long n_proposals = 0;
    int diverging = 0, turning = 0;

    for (int it = 0; it < max_treedepth; ++it) { /* nuts.py:209-213 */
        int direction = logbern(&cx, log(0.5)) * 2 - 1;
        /* Tree.extend: nuts.py:45-103 */
        subtree tree;
        /* the begin/end states of the two halves, as (p, v) pointers; copied because left/right get replaced */
        lf_state old_left = state_clone(&left, d), old_right = state_clone(&right, d);
        if (direction > 0) {
            tree = build_subtree(&cx, &right, depth, step_size, &diverging, &turning);
            if (tree.has) { state_free(&right); right = state_clone(&tree.right, d); }
        } else {
            tree = build_subtree(&cx, &left, depth, -step_size, &diverging, &turning);
            if (tree.has) { state_free(&left); left = state_clone(&tree.right, d); }
        }
        depth += 1;
        accept_sum += tree.accept_sum;
        n_proposals += tree.n_proposals;
        if (cx.err) { subtree_free(&tree); state_free(&old_left); state_free(&old_right); break; }
        if (!(diverging || turning)) {
            if (logbern(&cx, tree.log_size - log_size)) { /* nuts.py:81-83 */
                memcpy(prop_q, tree.prop_q, sizeof(double) * (size_t)d);
                prop_energy = tree.prop_energy; prop_logp = tree.prop_logp;
                prop_u = tree.prop_u; prop_weight = tree.prop_weight;
            }
            { /* nuts.py:85 */
                double a = log_size, b = tree.log_size, mx = a > b ? a : b, mn = a > b ? b : a;
                log_size = (mx == -INFINITY) ? -INFINITY : mx + log1p(exp(mn - mx));
            }
            for (int i = 0; i < d; ++i) p_sum[i] += tree.p_sum[i]; /* nuts.py:86, in place */
            /* nuts.py:90-101.  NOTE (reference behaviour, kept on purpose): `leftmost_p_sum = self.p_sum`
             * (direction > 0) and `rightmost_p_sum = self.p_sum` (direction < 0) alias the array that
             * line 86 has just updated in place, so those two carry the FULL new p_sum. */
            int turn = (dot(p_sum, left.v, d) <= 0) || (dot(p_sum, right.v, d) <= 0);
            const lf_state *lm_begin, *lm_end, *rm_begin, *rm_end;
            const double *lm_psum, *rm_psum;
            if (direction > 0) {
                lm_begin = &old_left; lm_end = &old_right; rm_begin = &tree.left; rm_end = &tree.right;
                lm_psum = p_sum; rm_psum = tree.p_sum;
            } else {
                lm_begin = &tree.right; lm_end = &tree.left; rm_begin = &old_left; rm_end = &old_right;
                lm_psum = tree.p_sum; rm_psum = p_sum;
            }
            for (int i = 0; i < d; ++i) tmp[i] = lm_psum[i] + rm_begin->p[i];
            int turn1 = (dot(tmp, lm_begin->v, d) <= 0) || (dot(tmp, rm_begin->v, d) <= 0);
            for (int i = 0; i < d; ++i) tmp[i] = lm_end->p[i] + rm_psum[i];
            int turn2 = (dot(tmp, lm_end->v, d) <= 0) || (dot(tmp, rm_end->v, d) <= 0);
            turning = turn | turn1 | turn2;
        }
        subtree_free(&tree);
        state_free(&old_left);
        state_free(&old_right);
        if (cx.err || diverging || turning) break;
    }
    int rc = cx.err;
    if (!rc) {
        double mean_tree_accept = accept_sum / (double)n_proposals; /* nuts.py:186 */
        step_size_update(c, mean_tree_accept, warmup);              /* base_hmc.py:80 */
        metric_update(c, prop_q, warmup);                           /* base_hmc.py:81 */
        memcpy(c->q, prop_q, sizeof(double) * (size_t)d);
        memcpy(sample_out, prop_q, sizeof(double) * (size_t)d);
        st[BFO_ST_LOGP] = prop_logp;
        st[BFO_ST_ENERGY] = prop_energy;
        st[BFO_ST_TREE_DEPTH] = depth;
        st[BFO_ST_TREE_SIZE] = (double)n_proposals;
        st[BFO_ST_MEAN_TREE_ACCEPT] = mean_tree_accept;
        st[BFO_ST_STEP_SIZE] = exp(c->log_step);  /* after the update: base_hmc.py:82-83 */
        st[BFO_ST_STEP_SIZE_BAR] = exp(c->log_bar);
        st[BFO_ST_WARMUP] = warmup;
        st[BFO_ST_ENERGY_CHANGE] = prop_energy - start.energy;
        st[BFO_ST_MAX_ENERGY_CHANGE] = cx.max_energy_change;
        st[BFO_ST_DIVERGING] = diverging;
        if (base) { *u_io = prop_u; st_t[0] = prop_u; st_t[1] = prop_weight; } /* tnuts.py:21-32 */
        c->i_iter += 1;
        *n_leapfrog += cx.n_leapfrog;
    }
    state_free(&start); state_free(&left); state_free(&right);
    free(prop_q);
    if (!rc && rng->exhausted) rc = -2;
    return rc;
}

static int nuts_iteration(const bfo_density *dn, bfo_chain *c, bfo_rng *rng, int warmup, int max_treedepth,
                          double max_change, double *sample_out, double *st, long *n_leapfrog) {
    return nuts_iteration_t(dn, c, rng, warmup, max_treedepth, max_change, sample_out, st, n_leapfrog, NULL, 0., NULL, NULL);
}

/* ================= tempered integrator: samplers/hmc_utils/integration.py:98-222 ================================ */
static double t_beta(double u) { return 1 / (1 + exp(-u)); }                                   /* :108-110 */
static double t_dbeta(double u) { double e = exp(-u); return e / ((1 + e) * (1 + e)); }         /* :113-116 */
static double t_pot(double u) { return u + 2 * log(1 + exp(-u)); }                              /* :119-124 */
static double t_dpot(double u) { double e = exp(u); return (e - 1) / (e + 1); }                 /* :127-130 */

/* phi, dphi = -logp_and_grad(q); psi, dpsi = -(base.logp_and_grad(q) + logxi): base_hmc.py:227-231 */
static void t_potentials(const bfo_density *dn, const bfo_density *base, double logxi, const double *q, double *phi, double *dphi,
                         double *psi, double *dpsi, int d) {
    bfo_logp_and_grad(dn, q, 0, phi, dphi);
    bfo_logp_and_grad(base, q, 0, psi, dpsi);
    *phi = -*phi;
    *psi = -(*psi + logxi);
    for (int i = 0; i < d; ++i) { dphi[i] = -dphi[i]; dpsi[i] = -dpsi[i]; }
}

/* compute_state: integration.py:132-151 (q, p, u, vt given) */
static void tempered_state(const bfo_density *dn, const bfo_density *base, double logxi, const bfo_chain *c, lf_state *s) {
    int d = c->d;
    double *dphi = (double *)malloc(sizeof(double) * (size_t)d * 2), *dpsi = dphi + d, phi, psi;
    t_potentials(dn, base, logxi, s->q, &phi, dphi, &psi, dpsi, d);
    metric_velocity(c, s->p, s->v);
    double kin = 0.;
    for (int i = 0; i < d; ++i) kin += s->p[i] * s->v[i];
    double kinetic = 0.5 * kin + s->vt * s->vt / 2;
    double beta = t_beta(s->u);
    s->energy = kinetic + (beta * phi + (1 - beta) * psi + t_pot(s->u));
    s->logp = -phi;
    double delta = phi - psi;
    s->weight = (delta == 0) ? 1. : delta / expm1(delta);
    free(dphi);
}

/* _step: integration.py:153-222 */
static void tempered_step(const bfo_density *dn, const bfo_density *base, double logxi, const bfo_chain *c, double eps,
                          const lf_state *s0, lf_state *s1) {
    int d = c->d;
    double dt = 0.5 * eps;
    double *dphi = (double *)malloc(sizeof(double) * (size_t)d * 2), *dpsi = dphi + d, phi, psi;
    double u = s0->u, vt = s0->vt;
    memcpy(s1->q, s0->q, sizeof(double) * (size_t)d);
    memcpy(s1->p, s0->p, sizeof(double) * (size_t)d);
    memcpy(s1->v, s0->v, sizeof(double) * (size_t)d);
    u += vt * dt;                                                        /* :173 */
    for (int i = 0; i < d; ++i) s1->q[i] += dt * s1->v[i];              /* :176 */
    t_potentials(dn, base, logxi, s1->q, &phi, dphi, &psi, dpsi, d);     /* :180-181 */
    double beta = t_beta(u), dbeta = t_dbeta(u), dU = t_dpot(u);
    double dpot_du = dbeta * (phi - psi) + dU;                            /* :185 */
    vt += -dpot_du * eps;                                                 /* :190 */
    for (int i = 0; i < d; ++i) s1->p[i] += eps * -(beta * dphi[i] + (1 - beta) * dpsi[i]); /* :186,193 */
    u += vt * dt;                                                         /* :198 */
    metric_velocity(c, s1->p, s1->v);                                     /* :201 */
    for (int i = 0; i < d; ++i) s1->q[i] += dt * s1->v[i];               /* :202 */
    double kin = 0.;
    for (int i = 0; i < d; ++i) kin += s1->p[i] * s1->v[i];
    double kinetic = 0.5 * kin + vt * vt / 2;                              /* :205-206 */
    t_potentials(dn, base, logxi, s1->q, &phi, dphi, &psi, dpsi, d);      /* :208-209 */
    beta = t_beta(u);
    s1->energy = (beta * phi + (1 - beta) * psi + t_pot(u)) + kinetic;    /* :210-213 */
    s1->logp = -phi;
    double delta = phi - psi;
    s1->weight = (delta == 0) ? 1. : delta / expm1(delta);               /* :219-220 */
    s1->u = u;
    s1->vt = vt;
    free(dphi);
}

/* TState after compute_state and after steps of the given sizes, for the fixtures: out rows of (2 d + 5): q, p, u, vt, weight, energy, logp */
void bfo_tempered_states(const bfo_density *dn, const bfo_density *base, double logxi, const double *var, const double *q0,
                         const double *p0, double u0, double v0, const double *eps, int n_step, double *out) {
    int d = dn->d;
    bfo_chain c;
    memset(&c, 0, sizeof(c));
    c.d = d; c.var = (double *)var;
    lf_state a = state_alloc(d), b = state_alloc(d);
    memcpy(a.q, q0, sizeof(double) * (size_t)d);
    memcpy(a.p, p0, sizeof(double) * (size_t)d);
    a.u = u0; a.vt = v0;
    tempered_state(dn, base, logxi, &c, &a);
    for (int k = 0; k <= n_step; ++k) {
        double *o = out + (size_t)k * (2 * d + 5);
        memcpy(o, a.q, sizeof(double) * (size_t)d);
        memcpy(o + d, a.p, sizeof(double) * (size_t)d);
        o[2 * d] = a.u; o[2 * d + 1] = a.vt; o[2 * d + 2] = a.weight; o[2 * d + 3] = a.energy; o[2 * d + 4] = a.logp;
        if (k == n_step) break;
        tempered_step(dn, base, logxi, &c, eps[k], &a, &b);
        lf_state t = a; a = b; b = t;
    }
    state_free(&a); state_free(&b);
}

/* TNUTS: BaseTHMC.astep (base_hmc.py:233-262) around the NUTS tree with the tempered integrator (tnuts.py).
 * u: in = u_0 of the first iteration (the reference draws it from numpy's global generator, base_hmc.py:241), out = last u.
 * stats: (n_run, BFO_N_NSTATS); stats_t: (n_run, 2) = u, weight of every sample. */
int bfo_tnuts_run(const bfo_density *dn, const bfo_density *base, double logxi, bfo_chain *c, bfo_rng *rng, double *u, long n_run,
                  long n_warmup, int max_treedepth, double max_change, double *samples, double *stats, double *stats_t) {
    long nl = 0;
    for (long i = 0; i < n_run; ++i) {
        int warmup = c->i_iter < n_warmup;
        int rc = nuts_iteration_t(dn, c, rng, warmup, max_treedepth, max_change, samples + (size_t)i * c->d,
                                  stats + (size_t)i * BFO_N_NSTATS, &nl, base, logxi, u, stats_t + (size_t)i * 2);
        if (rc) return rc;
    }
    return 0;
}

int bfo_nuts_run(const bfo_density *dn, bfo_chain *c, bfo_rng *rng, long n_run, long n_warmup,
                 int max_treedepth, double max_change, double *samples, double *stats) {
    long nl = 0;
    for (long i = 0; i < n_run; ++i) {
        int warmup = c->i_iter < n_warmup; /* base_hmc.py:155 */
        int rc = nuts_iteration(dn, c, rng, warmup, max_treedepth, max_change, samples + (size_t)i * c->d,
                                stats + (size_t)i * BFO_N_NSTATS, &nl);
        if (rc) return rc;
    }
    return 0;
}

/* samplers/hmc.py:16-49 inside base_hmc.py:62-85 */
int bfo_hmc_run(const bfo_density *dn, bfo_chain *c, bfo_rng *rng, long n_run, long n_warmup,
                int n_int_step, double max_change, double *samples, double *stats) {
    int d = c->d;
    lf_state start = state_alloc(d), a = state_alloc(d), b = state_alloc(d);
    int rc = 0;
    for (long it = 0; it < n_run && !rc; ++it) {
        int warmup = c->i_iter < n_warmup;
        double *st = stats + (size_t)it * BFO_N_HSTATS;
        metric_random(c, rng, start.p);
        memcpy(start.q, c->q, sizeof(double) * (size_t)d);
        bfo_logp_and_grad(dn, start.q, 0, &start.logp, start.grad);
        double kin = 0.;
        metric_velocity(c, start.p, start.v);
        for (int i = 0; i < d; ++i) kin += start.p[i] * start.v[i];
        start.energy = 0.5 * kin - start.logp;
        if (!isfinite(start.energy)) { rc = -1; break; }
        double step_size = exp(warmup ? c->log_step : c->log_bar);
        memcpy(a.q, start.q, sizeof(double) * (size_t)d * 4);
        a.energy = start.energy; a.logp = start.logp;
        lf_state *cur = &a, *nxt = &b;
        for (int s = 0; s < n_int_step; ++s) { /* hmc.py:19-20 */
            leapfrog_chain(dn, c, step_size, cur->q, cur->p, cur->grad, nxt->q, nxt->p, nxt->v, nxt->grad,
                         &nxt->energy, &nxt->logp);
            lf_state *t = cur; cur = nxt; nxt = t;
        }
        double energy_change;
        int diverging = 0;
        if (isfinite(cur->energy)) { /* hmc.py:21-28 */
            energy_change = start.energy - cur->energy;
            if (fabs(energy_change) > max_change) diverging = 1;
        } else { /* hmc.py:29-32 */
            energy_change = -INFINITY;
            diverging = 1;
        }
        double accept_stat = exp(energy_change);
        if (accept_stat > 1.) accept_stat = 1.;
        /* hmc.py:40-46: the uniform is only drawn when there is no divergence (short-circuit `or`) */
        int accepted = 0;
        if (!diverging) accepted = !(bfo_rng_uniform(rng) >= accept_stat);
        const lf_state *end = accepted ? cur : &start;
        step_size_update(c, accept_stat, warmup);
        metric_update(c, end->q, warmup);
        memcpy(c->q, end->q, sizeof(double) * (size_t)d);
        memcpy(samples + (size_t)it * d, end->q, sizeof(double) * (size_t)d);
        /* hmc.py:51-60: stats report the END-OF-TRAJECTORY state even when rejected */
        st[BFO_HS_LOGP] = cur->logp;
        st[BFO_HS_ENERGY] = cur->energy;
        st[BFO_HS_N_INT_STEP] = n_int_step;
        st[BFO_HS_ACCEPT_STAT] = accept_stat;
        st[BFO_HS_ACCEPTED] = accepted;
        st[BFO_HS_STEP_SIZE] = exp(c->log_step);
        st[BFO_HS_STEP_SIZE_BAR] = exp(c->log_bar);
        st[BFO_HS_WARMUP] = warmup;
        st[BFO_HS_ENERGY_CHANGE] = energy_change;
        st[BFO_HS_DIVERGING] = diverging;
        c->i_iter += 1;
        if (rng->exhausted) rc = -2;
    }
    state_free(&start); state_free(&a); state_free(&b);
    return rc;
}

long bfo_nuts_run_many(const bfo_density *dn, int n_chain, const double *x0, uint64_t seed,
                       uint64_t first_stream, long n_run, long n_warmup, int max_treedepth,
                       double max_change, double step_size, double target, int n_threads,
                       double *samples, double *stats) {
    int d = dn->d;
    long total = 0;
    int err = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int ci = 0; ci < n_chain; ++ci) {
        bfo_chain *c = bfo_chain_new(d, x0 + (size_t)ci * d, step_size, 1, target, 0.05, 0.75, 10., NULL, 1,
                                     NULL, 10., 60, 1, 1);
        bfo_rng rng;
        memset(&rng, 0, sizeof(rng));
        rng.kind = 0;
        bfo_xoshiro_seed(seed, first_stream + (uint64_t)ci, rng.s);
        double *smp = samples + (size_t)ci * n_run * d;
        double *st = stats + (size_t)ci * n_run * BFO_N_NSTATS;
        int rc = bfo_nuts_run(dn, c, &rng, n_run, n_warmup, max_treedepth, max_change, smp, st);
        if (rc) {
#pragma omp atomic write
            err = rc;
        } else {
            for (long i = 0; i < n_run; ++i) total += (long)st[(size_t)i * BFO_N_NSTATS + BFO_ST_TREE_SIZE];
        }
        bfo_chain_free(c);
    }
    return err ? (long)err : total;
}

/* Persistent-chain variant for the CPU baseline: chains[i] continue from their state; returns leapfrogs. */
long bfo_nuts_run_chains(const bfo_density *dn, int n_chain, bfo_chain **chains, bfo_rng *rngs, long n_run,
                         long n_warmup, int max_treedepth, double max_change, int n_threads, double *samples,
                         double *stats) {
    int d = dn->d;
    long total = 0;
    int err = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int ci = 0; ci < n_chain; ++ci) {
        double *smp = samples + (size_t)ci * n_run * d;
        double *st = stats + (size_t)ci * n_run * BFO_N_NSTATS;
        int rc = bfo_nuts_run(dn, chains[ci], &rngs[ci], n_run, n_warmup, max_treedepth, max_change, smp, st);
        if (rc) {
#pragma omp atomic write
            err = rc;
        } else {
            for (long i = 0; i < n_run; ++i) total += (long)st[(size_t)i * BFO_N_NSTATS + BFO_ST_TREE_SIZE];
        }
    }
    return err ? (long)err : total;
}

int bfo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ================= evidence path: utils/_cubic.pyx, utils/kde.py, evidence/bridge.py ======== */

/* utils/_cubic.pyx:24-96 (binary search with the same interval convention; the locality hint is not needed) */
static int find_interval(const double *x, int m, double v) {
    if (!(x[0] <= v && v < x[m - 1])) {
        if (v < x[0]) return 0;
        if (v >= x[m - 1]) return m;
        return -1;
    }
    int low = 1, high = m - 1;
    while (low < high) {
        int mid = (low + high) / 2;
        if (v < x[mid]) high = mid;
        else low = mid + 1;
    }
    return low;
}
static double cubic_eval(const double *c, double x) { return c[0] * x * x * x + c[1] * x * x + c[2] * x + c[3]; } /* :99-100 */
static double cubic_der(const double *c, double x) { return 3 * c[0] * x * x + 2 * c[1] * x + c[2]; }            /* :105-106 */

/* utils/_cubic.pyx:131-163 */
static double solve_bisect(const double *c, double yp, double x0, double x1) {
    const double tol = 1e-10;
    int i = 0;
    double a = 0., b = x1 - x0, x = (a + b) / 2, y = cubic_eval(c, x) - yp;
    while (!(y < tol && y > -tol)) {
        if (y > 0) b = x;
        else a = x;
        x = (a + b) / 2;
        y = cubic_eval(c, x) - yp;
        if (++i >= 100) { x = NAN; break; }
    }
    return x;
}

/* mode 0: evaluate (:188-231); 1: derivative (:237-279); 2: solve (:285-331).  c: (m+1, 4), x, y: (m,) */
void bfo_spline_apply(int mode, const double *c, const double *x, const double *y, int m, const double *in, double *out, size_t r) {
    for (size_t i = 0; i < r; ++i) {
        double v = in[i], o = NAN;
        int j = find_interval(mode == 2 ? y : x, m, v);
        if (mode == 0) {
            if (j > 0 && j < m) o = cubic_eval(c + 4 * j, v - x[j - 1]);
            else if (j == 0) o = c[2] * (v - x[0]) + c[3];
            else if (j == m) o = c[4 * m + 2] * (v - x[m - 1]) + c[4 * m + 3];
        } else if (mode == 1) {
            if (j > 0 && j < m) o = cubic_der(c + 4 * j, v - x[j - 1]);
            else if (j == 0) o = c[2];
            else if (j == m) o = c[4 * m + 2];
        } else {
            if (j > 0 && j < m) o = x[j - 1] + solve_bisect(c + 4 * j, v, x[j - 1], x[j]);
            else if (j == 0) o = x[0] + (v - c[3]) / c[2];
            else if (j == m) o = x[m - 1] + (v - c[4 * m + 3]) / c[4 * m + 2];
        }
        out[i] = o;
    }
}

/* kde.cdf for a 1-d KDE, utils/kde.py:322-354: sum_k w_k ndtr((x - data_k) / h) */
void bfo_kde_cdf(const double *data, const double *w, size_t n, double h, const double *pts, double *out, size_t m) {
    for (size_t i = 0; i < m; ++i) {
        double s = 0.;
        for (size_t k = 0; k < n; ++k) s += w[k] * 0.5 * erfc(-(pts[i] - data[k]) / h * 0.70710678118654752440);
        out[i] = s;
    }
}
