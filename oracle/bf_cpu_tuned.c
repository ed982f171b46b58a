/*
 * bf_cpu_tuned.c -- a tuned CPU evaluation of the common surrogate density, for bench.py's cpu_baseline leg ONLY
 * (TEST / MEASUREMENT INFRASTRUCTURE, like the rest of oracle/).
 *
 * The parity oracle (bf_oracle.c) follows the reference statement by statement: per-config gathers and scatters,
 * separate value and Jacobian passes over the upper-triangular coefficients, strided column reads, a heap allocation
 * per temporary.  That is the right checker and a weak baseline (SURVEY section 8d asks for the stronger one).  This
 * file evaluates the SAME density -- linear + quadratic configs over all inputs with the extrapolation bound, no
 * transform / scaling / decay (core/density.py:724-754, modules/poly.py:466-503) -- the way a CPU port would: one
 * symmetrised dense matrix S = A + A^T (value and gradient from one matvec), a second matvec for the bound test, no
 * allocation, unit-stride rows, compiled with -O3 -mavx2 -mfma.  Points outside the bound (rare) fall back to the
 * faithful path.  bfo_tuned_prepare() registers a density; the NUTS driver of bf_oracle.c then uses the tuned
 * evaluation through bfo_fast_hook.  Results agree with the faithful path to rounding (tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bf_oracle.h"

#define TUNED_MAX 8
#define TUNED_MAXD 128

typedef struct {
    const bfo_density *dn;
    int d;
    double *S;      /* (d,d) symmetric */
    double *lin;    /* (d,) */
    double c0;
} tuned_t;

static tuned_t g_tab[TUNED_MAX];
static int g_n = 0;

extern int (*bfo_fast_hook)(const bfo_density *, const double *, int, double *, double *);

static int tuned_eval(const bfo_density *dn, const double *x, int original_space, double *logp, double *grad) {
    (void)original_space; /* no transform: both spaces coincide */
    const tuned_t *t = NULL;
    for (int i = 0; i < g_n; ++i)
        if (g_tab[i].dn == dn) { t = &g_tab[i]; break; }
    if (!t) return 0;
    const int d = t->d;
    const bfo_poly_model *pm = &dn->poly;
    double xm[TUNED_MAXD], hv[TUNED_MAXD], g[TUNED_MAXD];
    double b2 = 0.;
    for (int i = 0; i < d; ++i) xm[i] = x[i] - pm->mu[i];
    for (int i = 0; i < d; ++i) {
        const double *row = pm->hess + (size_t)i * d;
        double s = 0.;
#pragma omp simd reduction(+ : s)
        for (int k = 0; k < d; ++k) s += row[k] * xm[k];
        hv[i] = s;
    }
    for (int i = 0; i < d; ++i) b2 += xm[i] * hv[i];
    if (!(b2 < pm->alpha * pm->alpha * (1. - 1e-12))) return 0; /* at or outside the bound: the faithful path decides */
    double f = t->c0;
    for (int i = 0; i < d; ++i) {
        const double *row = t->S + (size_t)i * d;
        double s = 0.;
#pragma omp simd reduction(+ : s)
        for (int k = 0; k < d; ++k) s += row[k] * x[k];
        g[i] = s + t->lin[i];
        f += x[i] * (t->lin[i] + 0.5 * s);
    }
    *logp = f;
    memcpy(grad, g, sizeof(double) * (size_t)d);
    return 1;
}

/* 0 on success, -1 when the density is not the common surrogate (nothing registered: the faithful path runs) */
int bfo_tuned_prepare(const bfo_density *dn) {
    const bfo_poly_model *pm = &dn->poly;
    const int d = dn->d;
    if (dn->ranges || dn->su_lo || dn->use_decay || dn->link_kind || !pm->use_bound || pm->n_config != 2 || d > TUNED_MAXD || g_n >= TUNED_MAX) return -1;
    const bfo_poly_config *cl = NULL, *cq = NULL;
    for (int c = 0; c < 2; ++c) {
        const bfo_poly_config *cf = &pm->configs[c];
        if (cf->n_in != d || cf->n_out != 1) return -1;
        for (int i = 0; i < d; ++i)
            if (cf->in_mask[i] != i) return -1;
        if (cf->order == BFO_LINEAR) cl = cf;
        else if (cf->order == BFO_QUADRATIC) cq = cf;
    }
    if (!cl || !cq) return -1;
    tuned_t *t = &g_tab[g_n];
    t->dn = dn;
    t->d = d;
    t->S = (double *)calloc((size_t)d * d, sizeof(double));
    t->lin = (double *)calloc((size_t)d, sizeof(double));
    t->c0 = cl->coef[0];
    for (int i = 0; i < d; ++i) t->lin[i] = cl->coef[1 + i];
    for (int j = 0; j < d; ++j)
        for (int k = j; k < d; ++k) { /* only j <= k is defined, modules/_poly.pyx:13-28 */
            const double a = cq->coef[(size_t)j * d + k];
            if (j == k) t->S[(size_t)j * d + j] = 2. * a;
            else { t->S[(size_t)j * d + k] = a; t->S[(size_t)k * d + j] = a; }
        }
    g_n += 1;
    bfo_fast_hook = tuned_eval;
    return 0;
}

void bfo_tuned_clear(void) {
    for (int i = 0; i < g_n; ++i) { free(g_tab[i].S); free(g_tab[i].lin); }
    g_n = 0;
    bfo_fast_hook = NULL;
}
