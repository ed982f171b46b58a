/*
 * bfhip.h -- C ABI of libbfhip.so: the MI355X (gfx950) implementation of BayesFast's data-parallel hot
 * path (polynomial-surrogate logp/grad, leapfrog, NUTS/HMC transitions over many chains, surrogate fit).
 *
 * The reference (h3jia/bayesfast) has no FFI boundary on this path: the seam is Python duck typing
 * (SURVEY.md section 8b).  Each entry point below therefore names the reference *Python/Cython interface*
 * it replaces (file:line relative to the reference root); INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C, no exceptions: every function returns 0 on success, <0 on error; bfhip_last_error()
 *     returns a message for the calling thread's last failure.
 *   - all array arguments of compute calls are DEVICE pointers owned by the caller (e.g. PyTorch-ROCm
 *     tensor.data_ptr()), row-major float64 unless stated; model descriptions are HOST pointers
 *     and are copied during the upload call.
 *   - one bfhip_ctx per (device, stream); calls on one ctx are not thread-safe, different ctxs are
 *     independent.  Kernels are launched asynchronously on the ctx's stream; nothing synchronises
 *     unless documented.
 */
#ifndef BFHIP_H
#define BFHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bfhip_ctx bfhip_ctx;

#define BFHIP_OK 0
#define BFHIP_ERR_ARG (-1)        /* invalid argument (the reference raises ValueError) */
#define BFHIP_ERR_STATE (-2)      /* bad state, e.g. no density uploaded (RuntimeError) */
#define BFHIP_ERR_HIP (-3)        /* HIP runtime failure */
#define BFHIP_ERR_UNSUPPORTED (-4)/* valid in the reference, not implemented on device yet (NotImplementedError) */

#define BFHIP_MAX_DIM 128         /* input_size limit of the device path */
#define BFHIP_MAX_TREEDEPTH 12

/* 102 (round 6): bfhip_polar_ns takes 2 d^2 + n_iter + 10 doubles of work.  101 (round 6): BFHIP_TREE_MODE_WORK grew to 4162 and
 * work[0] of bfhip_tree_size_mode_share carries the laggard bit; 100 before. */
int bfhip_version(void);
const char *bfhip_last_error(void);

/* stream: a hipStream_t (as void*), NULL for the default stream.  device: HIP ordinal. */
int bfhip_ctx_create(bfhip_ctx **out, int device, void *stream);
void bfhip_ctx_destroy(bfhip_ctx *ctx);
int bfhip_ctx_set_stream(bfhip_ctx *ctx, void *stream);
int bfhip_ctx_synchronize(bfhip_ctx *ctx);

/* ------------------------------------------------------------------------------------------------
 * Surrogate density description.  Replaces, as one flattened record, the objects walked per call by
 * Density.logp_and_grad (core/density.py:724-754): the Density's constraint transform
 * (core/density.py:92-140 -> transforms/_constraint.pyx:133-215), the Surrogate input scaling
 * (core/module.py:76-83,226), the PolyModel configs with their masks scattered to full input size
 * (modules/poly.py:466-478, modules/_poly.pyx:13-137), the linear-extrapolation bound
 * (modules/poly.py:480-503) and the decay penalty (core/density.py:740-746).
 * output_size of the surrogate is 1: the surrogate IS the log density -- or, with link_kind = 1, the input m of one
 * downstream analytic module, a Gaussian likelihood logp = link_logp0 - link_prec (m - link_y)^2 / 2, chained as the
 * pipeline does (core/density.py:527-560: the surrogate replaces the modules of its scope, the next module's Jacobian
 * multiplies the surrogate's; e.g. examples/2d-donut.ipynb: m = |x|, logp = -(m - 5)^2 / 0.5).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int d;                        /* input_size, 1..BFHIP_MAX_DIM */
    /* Density.input_scales (d,2) and hard_bounds (d,2) uint8; NULL => identity transform */
    const double *ranges;
    const uint8_t *hard_bounds;   /* may be NULL with ranges set => no hard bound */
    /* Surrogate.input_scales: lower edge (d,) and width (d,); NULL => none */
    const double *su_lo;
    const double *su_diff;
    /* polynomial in surrogate space, masks already scattered to the full input:
     *   f(x) = c0 + lin.x + sum_{j<=k} quad[j,k] x_j x_k + sum_{j,k} cubic2[j,k] x_j^2 x_k
     *          + sum_{j<k<l} cubic3[j,k,l] x_j x_k x_l                                              */
    double c0;
    const double *lin;            /* (d,)   or NULL */
    const double *quad;           /* (d,d)  only j<=k is read (the reference leaves the rest unset) */
    const double *cubic2;         /* (d,d)  or NULL */
    const double *cubic3;         /* (d,d,d) only j<k<l is read, or NULL */
    /* PolyModel bound, modules/poly.py:262-292 */
    int use_bound;
    const double *mu;             /* (d,) */
    const double *hess;           /* (d,d) */
    double alpha;
    double f_mu;
    /* Density decay, core/density.py:761-811.  When decay_mu / decay_hess hold the same numbers as mu / hess bit for bit -- the
     * reference takes both pairs from the fit points by the same statements (modules/poly.py:262-276, core/density.py:796-811) --
     * the decay term's product H_d (x - mu_d) is the bound's H (x - mu) and the kernels run it once. */
    int use_decay;
    const double *decay_mu;       /* (d,) */
    const double *decay_hess;     /* (d,d) */
    double decay_alpha2;
    double decay_gamma;
    /* downstream module of the surrogate's output (see above): 0 none, 1 Gaussian likelihood */
    int link_kind;
    double link_y, link_prec, link_logp0;
} bfhip_density_desc;

/* Copies and re-lays-out the description into device memory (MFMA A-operand fragments). Synchronous. */
int bfhip_density_upload(bfhip_ctx *ctx, const bfhip_density_desc *desc);

/* Density.logp_and_grad(x, original_space, use_surrogate=True) for n points.
 * x (n,d) -> logp (n,), grad (n,d).  Replaces core/density.py:724-754 (which loops rows in Python,
 * core/density.py:523-525).  grad may be NULL. */
int bfhip_logp_grad(bfhip_ctx *ctx, int n, const double *x, int original_space, double *logp, double *grad);

/* Constraint transforms of the uploaded density for n points, x (n,d) -> out (n,d):
 * which = 0 from_original, 1 from_original_grad, 2 from_original_grad2, 3 to_original, 4 to_original_grad,
 * 5 to_original_grad2  (Density.from_original/to_original/..., core/density.py:142-163 ->
 * transforms/_constraint.pyx:19-215; identity / ones / zeros when the density has no input_scales,
 * core/density.py:93-111).  bad (1,) int32 device flag: set to 1+i when variable i of some point is out of
 * bound in a from_original* call (the reference raises ValueError, _constraint.pyx:27-28). */
int bfhip_constraint(bfhip_ctx *ctx, int which, int n, const double *x, double *out, int *bad);

/* CpuLeapfrogIntegrator._step for n chains at once (samplers/hmc_utils/integration.py:68-95) with a
 * diagonal metric (QuadMetricDiag, samplers/hmc_utils/metrics.py:51-91).
 * eps (n,), var (n,d); q,p,grad (n,d) and logp,energy (n,) are updated in place; velocity_out (n,d) may be NULL. */
int bfhip_leapfrog(bfhip_ctx *ctx, int n, const double *eps, const double *var, double *q, double *p,
                   double *grad, double *logp, double *energy, double *velocity_out);

/* ------------------------------------------------------------------------------------------------
 * Sampler.  Per-chain state lives in caller-owned device arrays so that a later run() continues the
 * chains (SURVEY.md section 5 "checkpoint/resume"; samplers/hmc_utils/base_hmc.py:98-111).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    /* _HTrace / NTrace / HTrace hyper-parameters, samplers/sample_trace.py:159-172,460-512 */
    int sampler;                  /* 0 = NUTS (samplers/nuts.py), 1 = HMC (samplers/hmc.py) */
    int n_warmup;
    int max_treedepth;            /* NUTS, <= BFHIP_MAX_TREEDEPTH */
    int n_int_step;               /* HMC */
    double max_change;
    double target_accept, gamma, k, t_0;
    int adapt_step_size;
    int adapt_metric;
    int update_window;
    int doubling;
    /* Full-rank metric, QuadMetricFull / QuadMetricFullAdapt (samplers/hmc_utils/metrics.py:94-132,240-330).
     * 0 / NULL: the diagonal metric of `vec`.  Otherwise metric_mat points to the per-chain matrices filled by
     * bfhip_metric_init_full, (n_chain, BFHIP_MAT_N, d, d) doubles on the device; adapt_metric chooses between the
     * fixed (QuadMetricFull) and the adaptive (QuadMetricFullAdapt) variant exactly as for the diagonal metric. */
    int full_metric;
    double *metric_mat;
    /* How the chains of a 16-chain workgroup are laid out (common surrogate at d <= 64, diagonal metric; ignored elsewhere).
     * 1 = lane per chain (bfhip_group.hip): one instruction advances the scalar logic of all 16 chains; fastest while the
     *     chains of a workgroup stay in step (every tree the same size), slow when they do not, because every trip then
     *     executes the union of the chains' paths.
     * 2 = wave per chain (bfhip_sampler.hip: pipelined / sliced kernels): each chain follows its own path; insensitive to
     *     chains out of step.
     * 3 = "split": the lane-per-chain layout with eight waves per 16 chains, two per SIMD with disjoint work -- integrator
     *     waves (leapfrog step, gradient tiles) and bookkeeper waves (the NUTS tree, one leaf behind); NUTS on the plain
     *     common surrogate at 33 <= d <= 64, anything else runs as 1.  Bit-identical results to 1.
     * 0 = library default: 2 for NUTS, 1 for HMC (whose chains are always in step).  bayesfast_amd.chains.DeviceChains
     *     chooses 1 / 3 or 2 per run from the tree sizes of the previous run. */
    int chain_layout;
} bfhip_sampler_config;

/* matrices per chain for the full-rank metric (layout documented in bayesfast_amd/csrc/bfhip_metric.h; slot
 * BFHIP_MAT_COV holds the covariance transposed, i.e. as is for a symmetric matrix) */
#define BFHIP_MAT_COV 0
#define BFHIP_MAT_N 6

/* Layout of the per-chain scalar state array `sc` (C, BFHIP_SC_N) float64 */
enum {
    BFHIP_SC_LOG_STEP = 0,   /* DualAverageAdaptation._log_step, samplers/hmc_utils/step_size.py:13 */
    BFHIP_SC_LOG_BAR,        /* _log_bar */
    BFHIP_SC_HBAR,           /* _hbar */
    BFHIP_SC_MU,             /* _mu = log(10 * initial_step) */
    BFHIP_SC_COUNT,          /* _count */
    BFHIP_SC_FG_N,           /* _WeightedVariance.n_samples of the foreground window, metrics.py:337 */
    BFHIP_SC_BG_N,           /* same, background window */
    BFHIP_SC_N_SAMPLES,      /* QuadMetricDiagAdapt._n_samples */
    BFHIP_SC_PREV_UPDATE,    /* _previous_update */
    BFHIP_SC_ADAPT_WINDOW,   /* _adapt_window (doubles when `doubling`) */
    BFHIP_SC_I_ITER,         /* iterations done (len(trace._samples)) */
    BFHIP_SC_ERROR,          /* 0 ok; 1 bad initial energy (base_hmc.py:72-76); 2 logbern(NaN) (nuts.py:201-202); 3 metric covariance not positive definite (metrics.py:107-108) */
    BFHIP_SC_N
};

/* Layout of the per-chain vector state array `vec` (C, BFHIP_VEC_N, d) float64 */
enum {
    BFHIP_VEC_Q = 0,         /* current position: last sample, or x_0 (transformed space) */
    BFHIP_VEC_VAR,           /* QuadMetricDiag._var */
    BFHIP_VEC_FG_MEAN, BFHIP_VEC_FG_RAW, BFHIP_VEC_BG_MEAN, BFHIP_VEC_BG_RAW, /* metrics.py:333-371 */
    BFHIP_VEC_N
};

/* NUTS statistics per iteration, order of samplers/hmc_utils/stats.py:12-14 */
enum {
    BFHIP_NS_LOGP = 0, BFHIP_NS_ENERGY, BFHIP_NS_TREE_DEPTH, BFHIP_NS_TREE_SIZE, BFHIP_NS_MEAN_TREE_ACCEPT,
    BFHIP_NS_STEP_SIZE, BFHIP_NS_STEP_SIZE_BAR, BFHIP_NS_WARMUP, BFHIP_NS_ENERGY_CHANGE,
    BFHIP_NS_MAX_ENERGY_CHANGE, BFHIP_NS_DIVERGING, BFHIP_NS_N
};
/* HMC statistics, order of samplers/hmc_utils/stats.py:7-9 (one slot left unused to share the stride) */
enum {
    BFHIP_HS_LOGP = 0, BFHIP_HS_ENERGY, BFHIP_HS_N_INT_STEP, BFHIP_HS_ACCEPT_STAT, BFHIP_HS_ACCEPTED,
    BFHIP_HS_STEP_SIZE, BFHIP_HS_STEP_SIZE_BAR, BFHIP_HS_WARMUP, BFHIP_HS_ENERGY_CHANGE, BFHIP_HS_DIVERGING,
    BFHIP_HS_N
};
#define BFHIP_STAT_STRIDE 11

/* Runs every chain from its own i_iter up to (excluding) iter_end: BaseHMC.run/astep
 * (samplers/hmc_utils/base_hmc.py:62-85,155-156) with NUTS._hamiltonian_step (samplers/nuts.py:205-217,
 * Tree :21-189) or HMC._hamiltonian_step (samplers/hmc.py:16-49), for n_chain chains in one launch.
 *   rng     (C,4) uint64  per-chain xoshiro256++ state (bfhip_rng_seed)
 *   sc      (C,BFHIP_SC_N) float64, vec (C,BFHIP_VEC_N,d) float64: state above, updated in place
 *   samples (C,n_out,d): iteration i of a chain is written to row i - iter_out0
 *   stats   (C,n_out,BFHIP_STAT_STRIDE)
 *   n_leapfrog (1,) uint64 device counter, incremented by the number of leapfrog steps taken (may be NULL)
 * Asynchronous. */
int bfhip_sampler_run(bfhip_ctx *ctx, const bfhip_sampler_config *cfg, int n_chain, int iter_end,
                      uint64_t *rng, double *sc, double *vec, int iter_out0, int n_out, double *samples,
                      double *stats, unsigned long long *n_leapfrog);

/* Helper of the layout choice (chain_layout above): work[0] := the most common tree_size (1 .. 4095; larger sizes count as 4095)
 * when at least `share` of the NUTS trees in rows [row0, row0 + n_rows) of stats (C,n_out,BFHIP_STAT_STRIDE), all chains, have it,
 * else 0; plus 4096 -- whether or not the trees are in step -- when some chain's trees of these rows add up to at least four times
 * the mean over the chains (to the resolution of the size classes 1, 2, 3, 4, 6, 8, 12, ...: a launch lasts as long as its busiest
 * chain).  Read it as (work[0] & 4095, work[0] >> 12).
 * work: BFHIP_TREE_MODE_WORK int32 on the device, zeroed once by the caller and left clean by every call (the kernel writes all
 * BFHIP_TREE_MODE_WORK entries: a binding built against bfhip_version() 100, where the buffer was 4098 ints, must be rebuilt).
 * Queued on the context's stream behind the launch that wrote the rows; nothing synchronises. */
#define BFHIP_TREE_MODE_WORK 4162
int bfhip_tree_size_mode_share(bfhip_ctx *ctx, int n_chain, int n_out, const double *stats, int row0, int n_rows, double share,
                               int *work);

/* ------------------------------------------------------------------------------------------------
 * Tempered NUTS: TNUTS (samplers/tnuts.py:15-41) = BaseTHMC.astep
 * (samplers/hmc_utils/base_hmc.py:220-262) + the NUTS tree + TCpuLeapfrogIntegrator (samplers/hmc_utils/integration.py:98-222).
 * The target is the uploaded surrogate density (common surrogate: linear + quadratic configs with the bound, optionally behind the
 * constraint transform -- input_scales / hard bounds -- and with the decay term; surrogate input scales fold away at upload;
 * d <= 64, diagonal metric); the base density of TNTrace(density_base=..., logxi=...) (samplers/sample_trace.py:540-567,607-629) is a
 * quadratic log-density  c0 + lin.x + x.S x / 2  given by DEVICE arrays (e.g. a Gaussian).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const double *base_S;    /* (d,d) symmetric, device */
    const double *base_lin;  /* (d,), device */
    double base_c0;
    double logxi;            /* _TTrace.logxi */
} bfhip_tempering;

/* As bfhip_sampler_run (sampler = NUTS), plus: u (C,) the tempering coordinate of every chain (in: u_0 -- the reference draws
 * it from numpy's global generator, base_hmc.py:241 --, out: u of the last sample); stats_t (C,n_out,2) = (u, weight) of every
 * sample (TNStepStats, samplers/hmc_utils/stats.py).  One normal draw more per iteration than NUTS (v_0, base_hmc.py:245).
 * THMC is not offered: the reference's THTrace constructor raises (samplers/sample_trace.py:600). */
int bfhip_tnuts_run(bfhip_ctx *ctx, const bfhip_sampler_config *cfg, const bfhip_tempering *tp, int n_chain, int iter_end,
                    uint64_t *rng, double *sc, double *vec, double *u, int iter_out0, int n_out, double *samples, double *stats,
                    double *stats_t, unsigned long long *n_leapfrog);

/* Fills rng (C,4) with xoshiro256++ states for streams first_stream .. first_stream+C-1 of `seed`
 * (counterpart of utils/random.py:20-32 spawn_generator: one independent stream per GLOBAL chain index). */
int bfhip_rng_seed(bfhip_ctx *ctx, int n_chain, uint64_t seed, uint64_t first_stream, uint64_t *rng);

/* Initialises sc/vec for fresh chains: _HTrace._init_chain (samplers/sample_trace.py:178-202),
 * _set_step_size_2 (:365-373) and _set_metric_2 (:424-455).
 * x0 (C,d); metric_var (d,) or NULL (ones); initial_mean (d,) or NULL (x0 of each chain). */
int bfhip_chain_init(bfhip_ctx *ctx, int n_chain, int d, const double *x0, double step_size,
                     const double *metric_var, const double *initial_mean, double initial_weight,
                     int adapt_window, double *sc, double *vec);

/* Full-rank metric for fresh chains: _set_metric_2 with a 2-d metric (samplers/sample_trace.py:424-455),
 * QuadMetricFull.__init__ / QuadMetricFullAdapt.__init__ (metrics.py:103-111,261-285).
 * cov0 (d,d) row-major or NULL (identity), shared by all chains; mat (C, BFHIP_MAT_N, d, d).  Call after
 * bfhip_chain_init (the Welford means and weights live in sc/vec).  A cov0 that is not positive definite sets
 * sc[:, BFHIP_SC_ERROR] = 3 for every chain (the reference raises ValueError, metrics.py:107-108). */
int bfhip_metric_init_full(bfhip_ctx *ctx, int n_chain, int d, const double *cov0, double initial_weight,
                           double *sc, double *mat);

/* ------------------------------------------------------------------------------------------------
 * Multi-output surrogate module: PolyModel.fun / jac / fun_and_jac for output_size m > 1
 * (modules/poly.py:430-503).  Linear, quadratic and cubic configs, masks already scattered by
 * the caller: c0 (m), lin (m,d), quad (m,d,d) with the upper triangle j <= k as in bfhip_density_desc.
 * The extrapolation bound (mu, hess, alpha) is shared by all outputs, f_mu (m) is per output.
 * ---------------------------------------------------------------------------------------------- */
typedef struct bfhip_polymodel_desc {
    int d, m;
    const double *c0, *lin, *quad;  /* quad may be NULL (all-linear model) */
    int use_bound;
    const double *mu, *hess;        /* (d), (d,d) */
    double alpha;
    const double *f_mu;             /* (m) */
    /* cubic configs (modules/_poly.pyx:49-137), compact over the dimensions they touch: mask2 (n2,) / mask3 (n3,) sorted
     * dimension indices; cubic2 (m, n2, n2): f_o += sum_{j,k} cubic2[o,j,k] x_j^2 x_k; cubic3 (m, n3, n3, n3), only
     * j < k < l is read: f_o += sum_{j<k<l} cubic3[o,j,k,l] x_j x_k x_l (x restricted to the mask).  NULL / 0: none. */
    int n2;
    const int *mask2;
    const double *cubic2;
    int n3;
    const int *mask3;
    const double *cubic3;
} bfhip_polymodel_desc;

/* Copies the module into device memory owned by the context (host pointers in the descriptor). */
int bfhip_polymodel_upload(bfhip_ctx *ctx, const bfhip_polymodel_desc *desc);

/* f (n,m) and, if jac != NULL, jac (n,m,d) for n points x (n,d); device pointers.  Outside the bound every output
 * is extrapolated linearly from the projected point, PolyModel._fj_bound (modules/poly.py:480-503). */
int bfhip_polymodel_eval(bfhip_ctx *ctx, int n, const double *x, double *f, double *jac);

/* Downstream analytic module of a pipeline whose first module is the multi-output surrogate: a Gaussian likelihood
 * (chi-square) of its m outputs, with the pipeline's chain rule (core/density.py:552-560: jac = dot(J_out, J_in)) fused in:
 *   r = prec (f - y)  (prec (m,m) row-major, or prec_diag (m) when prec is NULL),  logp = -1/2 (f - y).r + logp0,
 *   grad = -J^T r,  J (n,m,d) the surrogate's Jacobians.
 * f (n,m), jac (n,m,d) as written by bfhip_polymodel_eval; logp (n), grad (n,d); grad may be NULL. */
int bfhip_chi2_stage(bfhip_ctx *ctx, int n, int m, int d, const double *f, const double *jac, const double *y, const double *prec,
                     const double *prec_diag, double logp0, double *logp, double *grad);

/* ------------------------------------------------------------------------------------------------
 * Pipeline density: the density of a Density whose module list is
 *   [model (replaced by a multi-output PolyModel surrogate), Gaussian likelihood of its m outputs, optional prior module]
 * with use_surrogate=True -- core/density.py:487-566 (the surrogate replaces the modules of its scope, :527-551; every
 * later module's Jacobian multiplies the ones before it, :552-560), modules/poly.py:430-503 (multi-output evaluation and
 * the bound's linear extrapolation), core/density.py:724-754 (transform, decay); the shape of
 * examples/des-y1-w-cosmosis.ipynb cells 12-18 (d = 27, m = 457: chi2_f / chi2_fj on the whitened data vector, des_post_f /
 * des_post_fj adding a Gaussian prior of some inputs).
 *   like  = logp0 - (f(x) - y)^T prec (f(x) - y) / 2          prec (m,m) symmetric positive definite, or diag(prec_diag)
 *   logp  = like + prior_c0 - sum_i prior_prec[i] (x_i - prior_mu[i])^2 / 2        (prior_*: original-space inputs; optional)
 * After the upload this IS the context's density: bfhip_logp_grad, bfhip_constraint and bfhip_sampler_run (NUTS / HMC,
 * diagonal metric, d <= 64) run on it.  The (m, d) Jacobian is never formed: the Cholesky factor of prec is folded into
 * the coefficient matrix at upload and grad = -(d phi / dx)^T C'^T (C' phi - y') is two FP64-MFMA contractions per
 * evaluation with the 16 chains of a workgroup as columns (bayesfast_amd/csrc/bfhip_pld.h).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int d, m;
    const double *ranges;         /* as bfhip_density_desc */
    const uint8_t *hard_bounds;
    const double *su_lo, *su_diff;
    bfhip_polymodel_desc model;   /* the surrogate: dense per-output coefficients, masks scattered; model.d = d, model.m = m */
    const double *y;              /* (m) */
    const double *prec;           /* (m,m) row-major, or NULL */
    const double *prec_diag;      /* (m) when prec is NULL */
    double logp0;
    const double *prior_mu;       /* (d) or NULL */
    const double *prior_prec;     /* (d) inverse variances, 0 where an input has no prior; NULL with prior_mu */
    double prior_c0;
    int use_decay;                /* as bfhip_density_desc */
    const double *decay_mu, *decay_hess;
    double decay_alpha2, decay_gamma;
} bfhip_pipeline_desc;

/* Copies, whitens and re-lays-out the description into device memory.  Synchronous.  BFHIP_ERR_ARG when prec is not positive
 * definite; BFHIP_ERR_UNSUPPORTED when d > 64 or the working set of one workgroup does not fit the CU's LDS. */
int bfhip_pipeline_upload(bfhip_ctx *ctx, const bfhip_pipeline_desc *desc);

/* ------------------------------------------------------------------------------------------------
 * Surrogate fit, PolyModel.fit (modules/poly.py:505-589).
 * ---------------------------------------------------------------------------------------------- */
/* Design-matrix block for one PolyConfig: x (n, n_in) gathered inputs -> A[:, col0 : col0+width] of the
 * row-major (n, lda) matrix A.  order: 0 linear (1 | x), 1 quadratic, 2 cubic-2, 3 cubic-3.
 * Replaces modules/poly.py:537-564 -> modules/_poly.pyx:143-177.  Optional row weights w (n,) scale the
 * block (modules/poly.py:566-568). */
int bfhip_design_block(bfhip_ctx *ctx, int order, int n, int n_in, const double *x, const double *w,
                       double *A, int lda, int col0);

/* Normal equations with FP64 MFMA: G (P,P) = A^T A, r (P,m) = A^T B for A (n,P), B (n,m), row-major.
 * Stands in for the factorisation inside scipy.linalg.lstsq (modules/poly.py:570). */
int bfhip_gram(bfhip_ctx *ctx, int n, int P, int m, const double *A, int lda, const double *B, double *G, double *r);

/* Solves G c = r for SPD G (P,P) in place by blocked Cholesky on device: r (P,m) is overwritten by the solution, G by
 * the factor of the Jacobi-equilibrated matrix in the solver's own layout (64 x 64 diagonal blocks of L in place, the
 * blocks below the diagonal TRANSPOSED in the upper triangle; the strict lower triangle holds work values).  info (1,)
 * int32 device flag: 0 ok, k>0 pivot k of the equilibrated matrix below 1e-11 (1-based, the first one). */
int bfhip_solve_spd(bfhip_ctx *ctx, int P, int m, double *G, double *r, int *info);

/* The least-squares solve of PolyModel.fit in one call (scipy.linalg.lstsq(A, b), modules/poly.py:570): c (P,m) =
 * argmin |A c - B| for A (n,lda>=P), B (n,m), by the normal equations above followed by n_refine steps of refinement on
 * the true residual, c += G^-1 A^T (B - A c), with the kept Cholesky factor (corrected semi-normal equations: the
 * coefficient error drops from cond(A)^2 eps to the cond(A) eps of an orthogonal factorisation).  G (P,P) receives the
 * factor; work holds n*m + P*m doubles (may be NULL when n_refine = 0).  info as in bfhip_solve_spd: a pivot of the
 * equilibrated matrix below 1e-11 is reported as rank deficiency and c is then not meaningful -- the caller
 * regularises, it is never silent.  Checked against the reference's gelsd up to a column-equilibrated cond(A) of 1e7
 * (two steps: 1e-10 of the coefficient scale, tests/golden/fit_illcond.npz). */
int bfhip_lstsq(bfhip_ctx *ctx, int n, int P, int m, const double *A, int lda, const double *B, double *G, double *c,
                int n_refine, double *work, int *info);

/* ------------------------------------------------------------------------------------------------
 * Refit glue: the steps either side of the sampler inside Recipe._sam_step / _pos_step.
 * ---------------------------------------------------------------------------------------------- */
/* Stable ascending sort of a (n,) float64 by value, every NaN last (the order of np.argsort(a), utils/misc.py:108, with
 * ties kept in index order): keys_sorted (n,) uint64 order-preserving keys, order (n,) int64 the permutation.
 * SystematicResampler.run(a, m) is order[ranks] for the m ranks of its index pattern (utils/misc.py:92-100). */
int bfhip_sort_keys(bfhip_ctx *ctx, long n, const double *a, uint64_t *keys_sorted, int64_t *order);

/* keys (n,) uint64 of a (n,) float64, unsorted (the same mapping as bfhip_sort_keys). */
int bfhip_order_keys(bfhip_ctx *ctx, long n, const double *a, uint64_t *keys);

/* For a sorted shard keys_sorted (n,): counts[i] = number of keys < q[i] (upper = 0) or <= q[i] (upper = 1), q (nq,).
 * The sharded form of the rank selection: ranks that share no data-path collective bisect on the key values and
 * all-reduce these counts (bayesfast_amd/core/refit.py; SURVEY section 8e option ii). */
int bfhip_count_keys(bfhip_ctx *ctx, long n, const uint64_t *keys_sorted, long nq, const uint64_t *q, int upper, int64_t *counts);

/* PostStep's importance weights (core/recipe.py:1289-1296): w = exp(logp - logq), w_trunc = clip(w, 0, mean(w) n^k_trunc)
 * (w_trunc = w for k_trunc < 0); all arrays (n,) float64 on the device. */
int bfhip_importance_weights(bfhip_ctx *ctx, long n, const double *logp, const double *logq, double k_trunc, double *w,
                             double *w_trunc);

/* ------------------------------------------------------------------------------------------------
 * Evidence: Gaussianized bridge sampling = SIT + bridge.
 * ---------------------------------------------------------------------------------------------- */
/* kde.cdf (utils/kde.py:322-354) of d one-dimensional weighted Gaussian KDEs at m points each:
 * out[j][i] = sum_k w[k] ndtr((pts[j][i] - data[j][k]) / h[j]).  data (d,n), w (n) normalised, h (d) bandwidths,
 * pts (d,m), out (d,m).  SIT._gaussianize_1d (transforms/sit.py:223-227) applies norm.ppf to it at the spline knots. */
int bfhip_kde_cdf(bfhip_ctx *ctx, int d, long n, const double *data, const double *w, const double *h, int m, const double *pts,
                  double *out);

/* The normal quantile function of n probabilities, out[i] = ndtri(p[i]) (scipy.special.ndtri, which scipy.stats.norm.ppf evaluates:
 * utils/sobol.py:57 on the Sobol points of multivariate_normal, transforms/sit.py:225 on the KDE cdfs): Cephes' algorithm; -inf / +inf
 * at 0 / 1, NaN outside [0, 1].  p and out may be the same array. */
int bfhip_ndtri(bfhip_ctx *ctx, long n, const double *p, double *out);

/* The Gaussianizing splines of one SIT iteration, all d coordinates in one launch (SIT._gaussianize_1d, transforms/sit.py:223-227:
 * cubic_spline(x, lambda xx: norm.ppf(kde.cdf(xx)), **cubic_options), utils/cubic.py:19-260).  sorted (d,n): every coordinate's
 * samples in ascending order (the percentiles); data (d,n), w (n) normalised, h (d): the KDE as in bfhip_kde_cdf.  grid (n_grid):
 * np.linspace(0, 100, bins + 1)[edge_bins:-edge_bins]; inner (n_inner): np.linspace(0, 100, edge_points + 2)[1:-1]; edge_bins,
 * max_width, split, max_add: cubic_spline's options.  Output per coordinate j: out_n[2 j] knots (0 when the coordinate was given up),
 * out_n[2 j + 1] flags (1 too few distinct knots / nothing beyond the edge knots, 2 singular slope system, 4 'the knots are too
 * unevenly spaced' (the reference raises), 8 more than 512 knots, 16 'Not all the intervals are monotone' (the reference warns));
 * knots out_x[j stride ..], values out_y[j stride ..], coefficient rows out_c[j 4 (stride + 1) ..] (m + 1 rows of 4, as
 * bfhip_spline_apply takes them).  stride >= 512, n_grid <= 512, n_inner <= 128. */
int bfhip_spline_build(bfhip_ctx *ctx, int d, long n, const double *sorted, const double *data, const double *w, const double *h,
                       int n_grid, const double *grid, int edge_bins, int n_inner, const double *inner, double max_width, int split,
                       int max_add, int stride, double *out_x, double *out_y, double *out_c, int *out_n);

/* The per-dimension piecewise cubics of SIT for n points x (n,d) -> out (n,d): mode 0 evaluate, 1 derivative, 2 solve
 * (utils/_cubic.pyx:188-336, called per dimension by SIT.forward_transform / backward_transform, transforms/sit.py:372-451).
 * Dimension j owns knots[knot_off[j] .. knot_off[j+1]), values there, and m_j + 1 coefficient rows of 4 at
 * coef[(knot_off[j] + j) * 4] (cubic_spline._x, ._y, ._c of utils/cubic.py).  knot_off (d+1,) int32. */
int bfhip_spline_apply(bfhip_ctx *ctx, int mode, long n, int d, const double *x, const int *knot_off, const double *knots,
                       const double *values, const double *coef, double *out);

/* Score function of the bridge estimator (evidence/bridge.py:44-49): out2[0] = logsumexp_i(logr + a_i - logaddexp(logr + a_i, 0)),
 * out2[1] = logsumexp_j(-logr + b_j - logaddexp(-logr + b_j, 0)); score = out2[0] - out2[1]. */
int bfhip_bridge_sums(bfhip_ctx *ctx, long n_a, const double *a, long n_b, const double *b, double logr, double *out2);

/* Per-sample terms of its error estimate (evidence/bridge.py:52-57): f1 (n_q), f2 (n_p). */
int bfhip_bridge_terms(bfhip_ctx *ctx, long n_p, const double *logp_p, const double *logq_p, long n_q, const double *logp_q,
                       const double *logq_q, double logr, double *f1, double *f2);

/* The glue of one FastICA iteration (scikit-learn's _ica_par, logcosh contrast, as SIT calls it: transforms/sit.py:235-244) around
 * the caller's two products Y = X1 W^T and P[b] = G_b^T X1_b (row batches b) and bfhip_polar_ns:
 *   bfhip_ica_tanh      y (n_pad,d) <- tanh(y) in place; partial (ceil(n_pad / 32), d) <- column sums of 1 - tanh(y)^2 over the
 *                       rows < n of every block of 32 rows (rows n .. n_pad: zero padding, not counted)
 *   bfhip_ica_assemble  a (d,d) <- (sum_b p[b]) / n - gmean[:, None] w,  gmean = column sums of partial / n;  p (nb,d,d);
 *                       *meas_k <- 0 (the slot bfhip_ica_post reduces into; may be NULL)
 *   bfhip_ica_post      meas[k] <- max(meas[k], max_i | |sum_j w1[i][j] w[i][j]| - 1 |), meas[n_meas + k] <- resid[0];
 *                       wbuf[k] <- w1; w <- w1 */
int bfhip_ica_tanh(bfhip_ctx *ctx, long n, long n_pad, int d, double *y, double *partial);
int bfhip_ica_assemble(bfhip_ctx *ctx, int d, int nb, const double *p, long n, long n_pad, const double *partial, const double *w,
                       double *a, double *meas_k);
int bfhip_ica_post(bfhip_ctx *ctx, int d, const double *w1, double *w, const double *resid, int k, int n_meas, double *wbuf,
                   double *meas);

/* FastICA's symmetric decorrelation W <- (W W^T)^{-1/2} W (sklearn.decomposition FastICA, as SIT calls it: transforms/sit.py:235-244;
 * scikit-learn takes an eigen-decomposition of W W^T on the host in every fixed-point iteration): the orthogonal polar factor of
 * a (d,d), by AT MOST n_iter Newton-Schulz steps on the FP64 matrix cores (the iteration stops once max |x x^T - I| < 1e-13), into
 * x (d,d).  work: 2 d^2 + n_iter + 10 doubles (the iterates' second buffer, the grid barrier's counter, a residual per step: all
 * in the caller's memory, so that a HIP graph holding the launch stays valid); resid (1,) receives max |x x^T - I| (device memory:
 * nothing synchronises; a caller checks it when it next looks at the device).  d <= 1024. */
int bfhip_polar_ns(bfhip_ctx *ctx, int d, const double *a, double *x, int n_iter, double *work, double *resid);

/* NOT part of this interface: the library's test and tuning switches (force a chain layout, a kernel form or a chains-per-workgroup
 * count; attach measurement buffers; run a launch in one part).  They have ONE entry point each for integers and for buffers,
 * declared with their keys in include/bfhip_debug.h; they never change what a call computes, and a binding has no use for them. */

#ifdef __cplusplus
}
#endif
#endif
