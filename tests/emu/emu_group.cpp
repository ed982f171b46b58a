// emu_group.cpp -- host emulation of the group sampler kernel (bayesfast_amd/csrc/bfhip_group.h).
// TEST INFRASTRUCTURE ONLY: compiled into tests/emu/_build/libbf_emu.so and loaded by tests/test_group_emu.py, which
// compares the kernel's control flow and arithmetic with the CPU oracle without a GPU.  Nothing in bayesfast_amd/
// loads it; the product path is the HIP build of the same header.
#include <ucontext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define BF_HOST_EMU 1
#include "bfhip_group.h"
#include "bfhip_split.h"
#include "bfhip_pack.h"

// the library's tuning switches as the host emulation sees them: the one the upload path reads, from the environment
BfTune &bf_tune() {
    static BfTune t = [] {
        BfTune u;
        memset(&u, 0, sizeof(u));
        const char *e = getenv("BFHIP_NO_PROOF_WEIGHTS");
        u.no_proof_weights = e ? atoi(e) : 0;
        return u;
    }();
    return t;
}

// ---------------------------------------------------------------------------------------------------------------
// cooperative fibres: one per lane of the workgroup
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct Coll {
    int arrived = 0;
    unsigned long gen = 0;
};
struct Emu {
    int n = 0, cur = 0, group = 0, n_done = 0;
    std::vector<ucontext_t> ctx;
    std::vector<char *> stacks;
    std::vector<char> done;
    ucontext_t main_ctx;
    Coll wave[16], grp;
    double xa[16][64], xb[16][64];
    int pred[16];
    long spins = 0;
    bool dead = false;
    void (*body)() = nullptr;
};
Emu *E = nullptr;

void emu_yield() {
    // round robin over the fibres that have not finished
    int nxt = E->cur;
    for (int k = 0; k < E->n; ++k) {
        nxt = (nxt + 1) % E->n;
        if (!E->done[nxt]) break;
    }
    if (nxt == E->cur) {  // nobody else can run: a collective that will never complete
        E->dead = true;
        swapcontext(&E->ctx[E->cur], &E->main_ctx);
    }
    if (++E->spins > 2000000000L) {
        E->dead = true;
        swapcontext(&E->ctx[E->cur], &E->main_ctx);
    }
    const int prev = E->cur;
    E->cur = nxt;
    swapcontext(&E->ctx[prev], &E->ctx[nxt]);
}

void emu_barrier(Coll &c, int n) {
    const unsigned long g = c.gen;
    if (++c.arrived == n) {
        c.arrived = 0;
        c.gen++;
        E->spins = 0;
    } else {
        while (c.gen == g) emu_yield();
    }
}

void fibre_main() {
    E->body();
    E->done[E->cur] = 1;
    E->n_done++;
    if (E->n_done == E->n) {
        setcontext(&E->main_ctx);
    }
    // hand over to a fibre that is still running; this one never resumes
    int nxt = E->cur;
    for (int k = 0; k < E->n; ++k) {
        nxt = (nxt + 1) % E->n;
        if (!E->done[nxt]) break;
    }
    E->cur = nxt;
    setcontext(&E->ctx[nxt]);
}

// runs body() on n fibres as workgroup `group`; returns false on a deadlock (divergent collective)
bool emu_run_group(int n, int group, void (*body)()) {
    Emu e;
    E = &e;
    e.n = n;
    e.group = group;
    e.body = body;
    e.ctx.resize(n);
    e.stacks.resize(n);
    e.done.assign(n, 0);
    const size_t STK = 256 * 1024;
    for (int i = 0; i < n; ++i) {
        e.stacks[i] = (char *)malloc(STK);
        getcontext(&e.ctx[i]);
        e.ctx[i].uc_stack.ss_sp = e.stacks[i];
        e.ctx[i].uc_stack.ss_size = STK;
        e.ctx[i].uc_link = &e.main_ctx;
        makecontext(&e.ctx[i], fibre_main, 0);
    }
    e.cur = 0;
    swapcontext(&e.main_ctx, &e.ctx[0]);
    for (int i = 0; i < n; ++i) free(e.stacks[i]);
    const bool ok = !e.dead && e.n_done == n;
    E = nullptr;
    return ok;
}
}  // namespace

int emu_tid() { return E->cur; }
int emu_group() { return E->group; }
void emu_sync() { emu_barrier(E->grp, E->n); }
bool emu_any(bool p) {
    const int w = E->cur >> 6;
    Coll &c = E->wave[w];
    if (c.arrived == 0) E->pred[w] = 0;
    if (p) E->pred[w] = 1;
    emu_barrier(c, 64);
    const bool r = E->pred[w] != 0;
    emu_barrier(c, 64);
    return r;
}
double emu_xor_add(double v, int mask) {
    const int w = E->cur >> 6, l = E->cur & 63;
    E->xa[w][l] = v;
    emu_barrier(E->wave[w], 64);
    const double o = E->xa[w][l ^ mask];
    emu_barrier(E->wave[w], 64);
    return v + o;
}
double emu_xor_get(double v, int mask) {
    const int w = E->cur >> 6, l = E->cur & 63;
    E->xa[w][l] = v;
    emu_barrier(E->wave[w], 64);
    const double o = E->xa[w][l ^ mask];
    emu_barrier(E->wave[w], 64);
    return o;
}
// lanes with the mask bit clear: a + a(l ^ mask); lanes with it set: b + b(l ^ mask)
double emu_pair_add(double a, double b, int mask) {
    const int w = E->cur >> 6, l = E->cur & 63;
    E->xa[w][l] = a;
    E->xb[w][l] = b;
    emu_barrier(E->wave[w], 64);
    const double r = (l & mask) ? b + E->xb[w][l ^ mask] : a + E->xa[w][l ^ mask];
    emu_barrier(E->wave[w], 64);
    return r;
}
bf_acc4 emu_mfma(double a, double b, bf_acc4 c) {
    const int w = E->cur >> 6, l = E->cur & 63;
    E->xa[w][l] = a;  // A[i = l & 15][k = l >> 4]
    E->xb[w][l] = b;  // B[k = l >> 4][n = l & 15]
    emu_barrier(E->wave[w], 64);
    const int n = l & 15, gq = l >> 4;
    bf_acc4 out;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + gq;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fma(E->xa[w][i + 16 * k], E->xb[w][n + 16 * k], acc);
        out[r] = acc;
    }
    emu_barrier(E->wave[w], 64);
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// entry points (host pointers everywhere)
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct Launch {
    DevModel m;
    SamplerArgs a;
    double *lds;
    int W, nuts, fs;
} L;

template <int W, bool NUTS, int FS>
void body_t() { bf_group_body<W, NUTS, FS>(L.m, L.a, L.lds); }
template <int W>
void body_split() { bf_split_body<true, W>(L.m, L.a, L.lds); }

template <int W>
void (*pick_body())() {
#ifdef BF_EMU_ASAN  // the sanitizer build instantiates what its two test cases run (NUTS, plain and decay + transform, d <= 16)
    if (W != 1 || !L.nuts) return nullptr;
    if (L.fs == 1) return body_t<1, true, 1>;
    if (L.fs == 7) return body_t<1, true, 7>;
    return nullptr;
#else
    if (L.nuts) {
        switch (L.fs) {
        case 1: return body_t<W, true, 1>;
        case 3: return body_t<W, true, 3>;
        case 5: return body_t<W, true, 5>;
        case 7: return body_t<W, true, 7>;
        }
    } else {
        switch (L.fs) {
        case 1: return body_t<W, false, 1>;
        case 3: return body_t<W, false, 3>;
        case 5: return body_t<W, false, 5>;
        case 7: return body_t<W, false, 7>;
        }
    }
    return nullptr;
#endif
}
}  // namespace

// the proven eigenvalue bound behind the kernels' bound proof (bfhip_pack.h), exposed for tests
extern "C" double bfemu_bound_lam_max(const double *hess, int d) { return bf_bound_lam_max(hess, d); }

extern "C" void bfemu_rng_seed(int n_chain, uint64_t seed, uint64_t first_stream, uint64_t *rng) {
    for (int c = 0; c < n_chain; ++c) bf_seed_state(seed, first_stream + (uint64_t)c, rng + (size_t)c * 4);
}

// _HTrace._init_chain (samplers/sample_trace.py:178-202,365-373,424-455): the same values as bf_chain_init_kernel
extern "C" void bfemu_chain_init(int n_chain, int d, const double *x0, double step_size, const double *metric_var,
                                 const double *initial_mean, double initial_weight, int adapt_window, double *sc, double *vec) {
    const double initial_step = step_size / pow((double)d, 0.25);
    for (int c = 0; c < n_chain; ++c) {
        double *s = sc + (size_t)c * BFHIP_SC_N;
        s[BFHIP_SC_LOG_STEP] = log(initial_step);
        s[BFHIP_SC_LOG_BAR] = log(initial_step);
        s[BFHIP_SC_HBAR] = 0.;
        s[BFHIP_SC_MU] = log(10. * initial_step);
        s[BFHIP_SC_COUNT] = 1.;
        s[BFHIP_SC_FG_N] = initial_weight;
        s[BFHIP_SC_BG_N] = 10.;
        s[BFHIP_SC_N_SAMPLES] = 0.;
        s[BFHIP_SC_PREV_UPDATE] = 0.;
        s[BFHIP_SC_ADAPT_WINDOW] = (double)adapt_window;
        s[BFHIP_SC_I_ITER] = 0.;
        s[BFHIP_SC_ERROR] = 0.;
        double *v = vec + (size_t)c * BFHIP_VEC_N * d;
        for (int i = 0; i < d; ++i) {
            const double x = x0[(size_t)c * d + i];
            const double var = metric_var ? metric_var[i] : 1.;
            v[BFHIP_VEC_Q * d + i] = x;
            v[BFHIP_VEC_VAR * d + i] = var;
            v[BFHIP_VEC_FG_MEAN * d + i] = initial_mean ? initial_mean[i] : x;
            v[BFHIP_VEC_FG_RAW * d + i] = var * initial_weight;
            v[BFHIP_VEC_BG_MEAN * d + i] = 0.;
            v[BFHIP_VEC_BG_RAW * d + i] = 0.;
        }
    }
}

// bfhip_sampler_run through the group kernel, on the host.  Returns 0, -1 (unsupported model), -2 (deadlock: a
// collective was called from divergent control flow).
extern "C" int bfemu_sampler_run(const bfhip_density_desc *ds, const bfhip_sampler_config *cfg, int n_chain, int iter_end,
                                 uint64_t *rng, double *sc, double *vec, int iter_out0, int n_out, double *samples,
                                 double *stats, unsigned long long *n_leapfrog) {
    if (!ds->quad || !ds->use_bound || ds->su_lo || ds->cubic2 || ds->cubic3 || ds->d > 64 || cfg->full_metric) return -1;
    std::vector<double> h;
    const int DP = bf_pack_density(ds, h);
    const size_t MAT = (size_t)DP * DP;
    DevModel &m = L.m;
    memset(&m, 0, sizeof(m));
    m.d = ds->d;
    m.DP = DP;
    m.has_transform = ds->ranges != NULL;
    m.has_quad = 1;
    m.use_bound = 1;
    m.use_decay = ds->use_decay != 0;
    m.pd = h.data();
    m.Sf = h.data() + (size_t)PD_N * DP;
    m.Hf = m.Sf + MAT;
    m.Hdf = m.Hf + MAT;
    m.c0 = ds->c0;
    m.alpha = ds->alpha;
    m.lam_max = ds->use_bound ? bf_bound_lam_max_weighted(ds->hess, ds->d, h.data() + (size_t)PD_HD * DP) : 0.;
    m.lam_max_d = ds->use_decay ? bf_bound_lam_max_weighted(ds->decay_hess, ds->d, h.data() + (size_t)PD_HDD * DP) : 0.;
    m.f_mu = ds->f_mu;
    m.f_poly_mu = bf_poly_at_mu(ds);
    m.inv_alpha = ds->use_bound ? 1. / ds->alpha : 0.;
    m.decay_alpha2 = ds->decay_alpha2;
    m.decay_gamma = ds->decay_gamma;
    const int W = DP / 16;
    L.W = W;
    L.nuts = cfg->sampler == 0;
    L.fs = 1 | (m.use_decay ? 2 : 0) | (m.has_transform ? 4 : 0);
    SamplerArgs &a = L.a;
    memset(&a, 0, sizeof(a));
    a.cfg = *cfg;
    a.n_chain = n_chain;
    a.iter_end = iter_end;
    a.iter_out0 = iter_out0;
    a.n_out = n_out;
    a.rng = rng;
    a.sc = sc;
    a.vec = vec;
    a.samples = samples;
    a.stats = stats;
    a.n_leapfrog = n_leapfrog;
    const int groups = (n_chain + 15) / 16;
    a.nslot = W == 4 ? GroupGeo<4>::scratch_slots() : (W == 2 ? GroupGeo<2>::scratch_slots() : GroupGeo<1>::scratch_slots());
    // chain_layout 3: the split layout (bfhip_split.h), eight waves per group, where it applies
    const bool split = cfg->chain_layout == 3 && L.nuts && L.fs == 1;
    const int split_slots = W == 4 ? SplitGeoT<4>::scratch_slots() : (W == 2 ? SplitGeoT<2>::scratch_slots() : SplitGeoT<1>::scratch_slots());
    if (split && split_slots > a.nslot) a.nslot = split_slots;
    std::vector<double> scratch((size_t)groups * 16 * a.nslot * DP, 0.);
    a.scratch = scratch.data();
    const size_t lds_n = split ? (W == 4 ? SplitGeoT<4>::lds_doubles() : (W == 2 ? SplitGeoT<2>::lds_doubles() : SplitGeoT<1>::lds_doubles()))
                               : (W == 4 ? GroupGeo<4>::lds_doubles(3) : (W == 2 ? GroupGeo<2>::lds_doubles(3) : GroupGeo<1>::lds_doubles(3)));
    std::vector<double> lds(lds_n);
    void (*body)() = split ? (W == 4 ? body_split<4> : (W == 2 ? body_split<2> : body_split<1>)) : (W == 4 ? pick_body<4>() : (W == 2 ? pick_body<2>() : pick_body<1>()));
    if (!body) return -1;
    for (int g = 0; g < groups; ++g) {
        // uninitialised shared memory: NaN patterns, so that a read of a never-written slot shows up
        for (size_t i = 0; i < lds_n; ++i) lds[i] = __builtin_nan("");
        L.lds = lds.data();
        if (!emu_run_group(split ? 128 * W : 64 * W, g, body)) return -2;
    }
    return 0;
}
