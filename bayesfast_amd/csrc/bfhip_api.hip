// bfhip_api.hip -- context, error reporting, density upload, RNG seeding, chain initialisation.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bfhip_common.h"
#include "bfhip_pack.h"

static thread_local char g_err[512] = "";

int bf_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- test / tuning switches (include/bfhip_debug.h, bfhip_tune.h) ----
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
BfTune &bf_tune() {
    static BfTune t = [] {
        BfTune u;
        memset(&u, 0, sizeof(u));
        const char *nk = getenv("BFHIP_NUTS_KERNEL");
        u.no_group = env_int("BFHIP_NO_GROUP", (nk && (!strcmp(nk, "sliced") || !strcmp(nk, "pipe"))) ? 1 : 0);
        u.no_pipe = env_int("BFHIP_NO_PIPE", (nk && !strcmp(nk, "sliced")) ? 1 : 0);
        u.no_plain = env_int("BFHIP_NO_PLAIN", 0);
        u.no_quad = env_int("BFHIP_NO_QUAD", 0);
        u.wave_cpg = env_int("BFHIP_WAVE_CPG", 0);
        u.tail_relaunch = env_int("BFHIP_TAIL_RELAUNCH", 1);
        u.tail_stop = env_int("BFHIP_TAIL_STOP", 4);
        u.tail_q = env_int("BFHIP_TAIL_Q", 3);
        u.tail_max = env_int("BFHIP_TAIL_MAX", 4);
        u.lone = env_int("BFHIP_LONE", 1);
        u.lone_form = env_int("BFHIP_LONE_FORM", -1);
        u.pld_waves = env_int("BFHIP_PLD_WAVES", 0);
        u.cubic_form = env_int("BFHIP_CUBIC_FORM", 0);
        u.cubic_loops = env_int("BFHIP_CUBIC_LOOPS", 0);
        u.gram_one_wave = env_int("BFHIP_GRAM_ONE_WAVE", 0);
        u.chol_one_panel = env_int("BFHIP_CHOL_ONE_PANEL", 0);
        u.no_vel_ahead = env_int("BFHIP_NO_VEL_AHEAD", 0);
        u.tnuts_wpb = env_int("BFHIP_TNUTS_WPB", 0);
        u.tnuts_generic = env_int("BFHIP_TNUTS_GENERIC", 0);
        u.no_bound_proof = env_int("BFHIP_NO_BOUND_PROOF", 0);
        u.no_proof_weights = env_int("BFHIP_NO_PROOF_WEIGHTS", 0);
        u.pld_no_compress = env_int("BFHIP_PLD_NO_COMPRESS", 0);
        u.pld_no_cl = env_int("BFHIP_PLD_NO_CL", 0);
        u.no_decay_shared = env_int("BFHIP_NO_DECAY_SHARED", 0);
        u.polar_tiles = env_int("BFHIP_POLAR_TILES", 0);
        u.no_group_pld = env_int("BFHIP_NO_GROUP_PLD", 0);
        return u;
    }();
    return t;
}
static int *tune_field(const char *key) {
    BfTune &t = bf_tune();
    struct { const char *k; int *p; } tab[] = {
        {"no_group", &t.no_group}, {"no_pipe", &t.no_pipe}, {"no_plain", &t.no_plain}, {"no_quad", &t.no_quad}, {"wave_cpg", &t.wave_cpg},
        {"tail_relaunch", &t.tail_relaunch}, {"tail_stop", &t.tail_stop}, {"tail_q", &t.tail_q}, {"tail_max", &t.tail_max}, {"lone", &t.lone}, {"lone_form", &t.lone_form},
        {"pld_waves", &t.pld_waves}, {"cubic_form", &t.cubic_form}, {"cubic_loops", &t.cubic_loops}, {"gram_one_wave", &t.gram_one_wave}, {"chol_one_panel", &t.chol_one_panel}, {"no_vel_ahead", &t.no_vel_ahead}, {"tnuts_wpb", &t.tnuts_wpb}, {"tnuts_generic", &t.tnuts_generic}, {"no_bound_proof", &t.no_bound_proof},
        {"no_proof_weights", &t.no_proof_weights}, {"pld_no_compress", &t.pld_no_compress}, {"pld_no_cl", &t.pld_no_cl}, {"no_decay_shared", &t.no_decay_shared}, {"polar_tiles", &t.polar_tiles}, {"no_group_pld", &t.no_group_pld}};
    for (auto &e : tab)
        if (key && !strcmp(key, e.k)) return e.p;
    return NULL;
}
extern "C" int bfhip_debug_set(const char *key, long long value) {
    int *f = tune_field(key);
    if (!f) return bf_set_error(BFHIP_ERR_ARG, "bfhip_debug_set: unknown switch '%s'", key ? key : "(null)");
    *f = (int)value;
    return 0;
}
extern "C" long long bfhip_debug_get(const char *key) {
    const int *f = tune_field(key);
    return f ? *f : 0;
}
extern "C" int bfhip_debug_buffer(const char *key, void *device_ptr) {
    BfTune &t = bf_tune();
    unsigned long long *p = (unsigned long long *)device_ptr;
    if (key && !strcmp(key, "stamps")) t.stamps = p;
    else if (key && !strcmp(key, "stamps_lone")) t.stamps_lone = p;
    else if (key && !strcmp(key, "gstamps")) t.gstamps = p;
    else if (key && !strcmp(key, "group_counters")) t.group_counters = p;
    else return bf_set_error(BFHIP_ERR_ARG, "bfhip_debug_buffer: unknown buffer '%s'", key ? key : "(null)");
    return 0;
}
extern "C" const char *bfhip_debug_last_kernel(void) { return bf_tune().last_kernel; }

extern "C" int bfhip_version(void) { return 102; }   // 102: bfhip_polar_ns work = 2 d^2 + n_iter + 10 doubles; 101: BFHIP_TREE_MODE_WORK 4162 (was 4098), work[0] = size | laggard << 12
extern "C" const char *bfhip_last_error(void) { return g_err; }

extern "C" int bfhip_ctx_create(bfhip_ctx **out, int device, void *stream) {
    if (!out) return bf_set_error(BFHIP_ERR_ARG, "bfhip_ctx_create: out is NULL");
    int n_dev = 0;
    BF_HIP_CHECK(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return bf_set_error(BFHIP_ERR_ARG, "bfhip_ctx_create: device %d of %d", device, n_dev);
    BF_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    BF_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip is built for gfx950 only, found %s", prop.gcnArchName);
    bfhip_ctx *c = (bfhip_ctx *)calloc(1, sizeof(bfhip_ctx));
    c->device = device;
    c->stream = (hipStream_t)stream;
    c->n_cu = prop.multiProcessorCount;
    *out = c;
    return 0;
}

extern "C" void bfhip_ctx_destroy(bfhip_ctx *ctx) {
    if (!ctx) return;
    if (ctx->model_buf) (void)hipFree(ctx->model_buf);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->cubic_buf) (void)hipFree(ctx->cubic_buf);
    if (ctx->pm_buf) (void)hipFree(ctx->pm_buf);
    if (ctx->pld_buf) (void)hipFree(ctx->pld_buf);
    if (ctx->tail_buf) (void)hipFree(ctx->tail_buf);
    if (ctx->flow) (void)hipFree(ctx->flow);
    free(ctx);
}

extern "C" int bfhip_ctx_set_stream(bfhip_ctx *ctx, void *stream) {
    if (!ctx) return bf_set_error(BFHIP_ERR_ARG, "ctx is NULL");
    ctx->stream = (hipStream_t)stream;
    return 0;
}

extern "C" int bfhip_ctx_synchronize(bfhip_ctx *ctx) {
    if (!ctx) return bf_set_error(BFHIP_ERR_ARG, "ctx is NULL");
    BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int bfhip_density_upload(bfhip_ctx *ctx, const bfhip_density_desc *ds) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || !ds) return bf_set_error(BFHIP_ERR_ARG, "bfhip_density_upload: NULL argument");
    const int d = ds->d;
    if (d < 1 || d > BFHIP_MAX_DIM) return bf_set_error(BFHIP_ERR_ARG, "bfhip_density_upload: d = %d out of [1, %d]", d, BFHIP_MAX_DIM);
    if (ds->use_bound && (!ds->mu || !ds->hess || !(ds->alpha > 0.)))
        return bf_set_error(BFHIP_ERR_ARG, "use_bound needs mu, hess and alpha > 0");
    if (ds->use_decay && (!ds->decay_mu || !ds->decay_hess))
        return bf_set_error(BFHIP_ERR_ARG, "use_decay needs decay_mu and decay_hess");
    if ((ds->su_lo == NULL) != (ds->su_diff == NULL)) return bf_set_error(BFHIP_ERR_ARG, "su_lo and su_diff go together");
    std::vector<double> h;
    const int DP = bf_pack_density(ds, h);
    const size_t MAT = (size_t)DP * DP, n_dbl = h.size();
    BF_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = n_dbl * sizeof(double);
    if (ctx->model_bytes < bytes) {
        if (ctx->model_buf) BF_HIP_CHECK(hipFree(ctx->model_buf));
        ctx->model_buf = NULL;
        ctx->model_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->model_buf, bytes));
        ctx->model_bytes = bytes;
    }
    BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // nobody may still read the old model
    BF_HIP_CHECK(hipMemcpy(ctx->model_buf, h.data(), bytes, hipMemcpyHostToDevice));
    DevModel &m = ctx->model;
    memset(&m, 0, sizeof(m));
    m.d = d;
    m.DP = DP;
    m.has_transform = ds->ranges != NULL;
    m.has_su = ds->su_lo != NULL;
    m.has_quad = ds->quad != NULL;
    m.use_bound = ds->use_bound != 0;
    m.use_decay = ds->use_decay != 0;
    m.decay_shared = ds->use_bound && ds->use_decay && memcmp(ds->decay_hess, ds->hess, (size_t)d * d * sizeof(double)) == 0 &&
                     memcmp(ds->decay_mu, ds->mu, (size_t)d * sizeof(double)) == 0;
    const double *base = (const double *)ctx->model_buf;
    m.pd = base;
    m.Sf = base + (size_t)PD_N * DP;
    m.Hf = m.Sf + MAT;
    m.Hdf = m.Hf + MAT;
    // (decay_shared: EVERY kernel takes the decay term's product from the bound's fragments -- H (x - mu) where the reference writes
    // (x - mu) H, core/density.py:745: the same numbers for a symmetric H, and inv(cov) is symmetric to rounding -- so that the
    // two-matrix form of the pipelined kernel and the three-matrix forms of the others agree bit for bit)
    if (m.decay_shared) m.Hdf = m.Hf;
    m.c0 = ds->c0;
    m.alpha = ds->alpha;
    m.lam_max = ds->use_bound ? bf_bound_lam_max_weighted(ds->hess, ds->d, h.data() + (size_t)PD_HD * DP) : 0.;
    m.lam_max_d = ds->use_decay ? bf_bound_lam_max_weighted(ds->decay_hess, ds->d, h.data() + (size_t)PD_HDD * DP) : 0.;
    m.f_mu = ds->f_mu;
    m.f_poly_mu = bf_poly_at_mu(ds);
    m.inv_alpha = ds->use_bound ? 1. / ds->alpha : 0.;
    m.decay_alpha2 = ds->decay_alpha2;
    m.decay_gamma = ds->decay_gamma;
    if (ds->link_kind != 0 && ds->link_kind != 1) return bf_set_error(BFHIP_ERR_ARG, "bfhip_density_upload: unknown link_kind %d", ds->link_kind);
    m.has_link = ds->link_kind == 1;
    m.link_y = ds->link_y;
    m.link_prec = ds->link_prec;
    m.link_logp0 = ds->link_logp0;
    // ---- cubic terms, compact over the dimensions they touch (modules/_poly.pyx:49-137) ----
    m.has_cubic = (ds->cubic2 || ds->cubic3) ? 1 : 0;
    if (m.has_cubic) {
        std::vector<int> mask2, mask3, pos2(DP, -1), pos3(DP, -1);
        if (ds->cubic2) {
            for (int i = 0; i < d; ++i) {
                bool used = false;
                for (int k = 0; k < d && !used; ++k) used = ds->cubic2[(size_t)i * d + k] != 0. || ds->cubic2[(size_t)k * d + i] != 0.;
                if (used) { pos2[i] = (int)mask2.size(); mask2.push_back(i); }
            }
        }
        if (ds->cubic3) {
            std::vector<char> used(d, 0);
            for (int j = 0; j < d; ++j)
                for (int k = j + 1; k < d; ++k)
                    for (int l = k + 1; l < d; ++l)
                        if (ds->cubic3[((size_t)j * d + k) * d + l] != 0.) used[j] = used[k] = used[l] = 1;
            for (int i = 0; i < d; ++i)
                if (used[i]) { pos3[i] = (int)mask3.size(); mask3.push_back(i); }
        }
        const int n2 = (int)mask2.size(), n3 = (int)mask3.size();
        std::vector<double> A2((size_t)n2 * n2), A2t((size_t)n2 * n2), T3((size_t)n3 * n3 * n3, 0.);
        for (int a = 0; a < n2; ++a)
            for (int b = 0; b < n2; ++b) {
                A2[(size_t)a * n2 + b] = ds->cubic2[(size_t)mask2[a] * d + mask2[b]];
                A2t[(size_t)b * n2 + a] = A2[(size_t)a * n2 + b];
            }
        for (int a = 0; a < n3; ++a)
            for (int b = a + 1; b < n3; ++b)
                for (int c = b + 1; c < n3; ++c) {
                    const double v = ds->cubic3[((size_t)mask3[a] * d + mask3[b]) * d + mask3[c]];
                    const int p[3] = {a, b, c};
                    static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
                    for (int q = 0; q < 6; ++q)  // T3t[k][l][j]
                        T3[((size_t)p[perm[q][0]] * n3 + p[perm[q][1]]) * n3 + p[perm[q][2]]] = v;
                }
        const size_t ib = (size_t)(n2 + n3 + 2 * DP) * sizeof(int), ibp = (ib + 7) / 8 * 8;
        const size_t cbytes = ibp + (A2.size() * 2 + T3.size()) * sizeof(double) + 8;
        if (ctx->cubic_bytes < cbytes) {
            if (ctx->cubic_buf) BF_HIP_CHECK(hipFree(ctx->cubic_buf));
            ctx->cubic_buf = NULL;
            ctx->cubic_bytes = 0;
            BF_HIP_CHECK(hipMalloc(&ctx->cubic_buf, cbytes));
            ctx->cubic_bytes = cbytes;
        }
        std::vector<char> hb(cbytes, 0);
        int *ip = (int *)hb.data();
        memcpy(ip, mask2.data(), n2 * sizeof(int));
        memcpy(ip + n2, pos2.data(), DP * sizeof(int));
        memcpy(ip + n2 + DP, mask3.data(), n3 * sizeof(int));
        memcpy(ip + n2 + DP + n3, pos3.data(), DP * sizeof(int));
        double *dp = (double *)(hb.data() + ibp);
        memcpy(dp, A2.data(), A2.size() * sizeof(double));
        memcpy(dp + A2.size(), A2t.data(), A2t.size() * sizeof(double));
        memcpy(dp + 2 * A2.size(), T3.data(), T3.size() * sizeof(double));
        BF_HIP_CHECK(hipMemcpy(ctx->cubic_buf, hb.data(), cbytes, hipMemcpyHostToDevice));
        const int *dip = (const int *)ctx->cubic_buf;
        const double *ddp = (const double *)((const char *)ctx->cubic_buf + ibp);
        m.n2 = n2; m.n3 = n3;
        m.mask2 = dip; m.pos2 = dip + n2; m.mask3 = dip + n2 + DP; m.pos3 = dip + n2 + DP + n3;
        m.A2 = ddp; m.A2t = ddp + A2.size(); m.T3t = ddp + 2 * A2.size();
    }
    ctx->has_model = 1;
    return 0;
}

extern "C" int bfhip_rng_seed(bfhip_ctx *ctx, int n_chain, uint64_t seed, uint64_t first_stream, uint64_t *rng) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n_chain < 0 || !rng) return bf_set_error(BFHIP_ERR_ARG, "bfhip_rng_seed: invalid argument");
    std::vector<uint64_t> h((size_t)n_chain * 4);
    for (int c = 0; c < n_chain; ++c) bf_seed_state(seed, first_stream + (uint64_t)c, &h[(size_t)c * 4]);
    BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    BF_HIP_CHECK(hipMemcpy(rng, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    return 0;
}

// _HTrace._init_chain: samplers/sample_trace.py:178-202, :365-373 (step size), :424-455 (metric)
__global__ void bf_chain_init_kernel(int n_chain, int d, const double *__restrict__ x0, double log_step0, double mu0,
                                     const double *__restrict__ metric_var, const double *__restrict__ initial_mean,
                                     double initial_weight, int adapt_window, double *__restrict__ sc,
                                     double *__restrict__ vec) {
    const int c = blockIdx.x;
    if (c >= n_chain) return;
    double *s = sc + (size_t)c * BFHIP_SC_N;
    if (threadIdx.x == 0) {
        s[BFHIP_SC_LOG_STEP] = log_step0;   // step_size.py:13
        s[BFHIP_SC_LOG_BAR] = log_step0;    // :14
        s[BFHIP_SC_HBAR] = 0.;
        s[BFHIP_SC_MU] = mu0;               // :20
        s[BFHIP_SC_COUNT] = 1.;             // :19
        s[BFHIP_SC_FG_N] = initial_weight;  // metrics.py:337
        s[BFHIP_SC_BG_N] = 10.;             // _WeightedVariance(n) default initial_weight, metrics.py:335
        s[BFHIP_SC_N_SAMPLES] = 0.;
        s[BFHIP_SC_PREV_UPDATE] = 0.;
        s[BFHIP_SC_ADAPT_WINDOW] = (double)adapt_window;
        s[BFHIP_SC_I_ITER] = 0.;
        s[BFHIP_SC_ERROR] = 0.;
    }
    double *v = vec + (size_t)c * BFHIP_VEC_N * d;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const double x = x0[(size_t)c * d + i];
        const double var = metric_var ? metric_var[i] : 1.;
        v[BFHIP_VEC_Q * d + i] = x;
        v[BFHIP_VEC_VAR * d + i] = var;
        v[BFHIP_VEC_FG_MEAN * d + i] = initial_mean ? initial_mean[i] : x;  // sample_trace.py:436-437
        v[BFHIP_VEC_FG_RAW * d + i] = var * initial_weight;                 // metrics.py:347
        v[BFHIP_VEC_BG_MEAN * d + i] = 0.;
        v[BFHIP_VEC_BG_RAW * d + i] = 0.;
    }
}

extern "C" int bfhip_chain_init(bfhip_ctx *ctx, int n_chain, int d, const double *x0, double step_size,
                                const double *metric_var, const double *initial_mean, double initial_weight,
                                int adapt_window, double *sc, double *vec) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n_chain < 0 || d < 1 || d > BFHIP_MAX_DIM || !x0 || !sc || !vec || !(step_size > 0.) ||
        !(initial_weight > 0.) || adapt_window < 1)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_chain_init: invalid argument");
    if (n_chain == 0) return 0;
    const double initial_step = step_size / pow((double)d, 0.25);  // sample_trace.py:371-373
    hipLaunchKernelGGL(bf_chain_init_kernel, dim3(n_chain), dim3(64), 0, ctx->stream, n_chain, d, x0,
                       log(initial_step), log(10. * initial_step), metric_var, initial_mean, initial_weight,
                       adapt_window, sc, vec);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
