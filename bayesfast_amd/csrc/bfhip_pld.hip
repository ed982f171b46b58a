// bfhip_pld.hip -- pipeline density (bfhip_pld.h): upload (monomial table, whitening, MFMA fragments, gradient table) and
// the stand-alone batched logp / grad; the fused sampler's instantiation for it lives in bfhip_sampler.hip.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bfhip_common.h"
#include "bfhip_pack.h"
#include "bfhip_eval.h"
#include "bfhip_pld.h"
#include "bfhip_sampler_defs.h"

namespace {

struct Mono { int i[3]; };   // x_i0 x_i1 x_i2, index DP = the constant one

// prec (m,m) = L L^T, L lower triangular row-major; false when a pivot is not positive
bool cholesky_lower(const double *A, int m, std::vector<double> &L) {
    L.assign((size_t)m * m, 0.);
    for (int i = 0; i < m; ++i)
        for (int k = 0; k <= i; ++k) {
            double s = 0.5 * (A[(size_t)i * m + k] + A[(size_t)k * m + i]);
            for (int q = 0; q < k; ++q) s -= L[(size_t)i * m + q] * L[(size_t)k * m + q];
            if (i == k) {
                if (!(s > 0.) || !std::isfinite(s)) return false;
                L[(size_t)i * m + i] = std::sqrt(s);
            } else {
                L[(size_t)i * m + k] = s / L[(size_t)k * m + k];
            }
        }
    return true;
}

int roundup(int v, int q) { return (v + q - 1) / q * q; }

}  // namespace

extern "C" int bfhip_pipeline_upload(bfhip_ctx *ctx, const bfhip_pipeline_desc *ds) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || !ds) return bf_set_error(BFHIP_ERR_ARG, "bfhip_pipeline_upload: NULL argument");
    const int d = ds->d, m = ds->m;
    const bfhip_polymodel_desc &pm = ds->model;
    if (d < 1 || m < 1 || pm.d != d || pm.m != m) return bf_set_error(BFHIP_ERR_ARG, "bfhip_pipeline_upload: d, m and the model's disagree");
    if (d > BFHIP_MAX_DIM) return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_pipeline_upload: input_size %d > %d is not implemented", d, BFHIP_MAX_DIM);
    if (!pm.c0 || !pm.lin || !ds->y) return bf_set_error(BFHIP_ERR_ARG, "bfhip_pipeline_upload: c0, lin and y are required");
    if ((ds->prec == NULL) == (ds->prec_diag == NULL)) return bf_set_error(BFHIP_ERR_ARG, "give exactly one of prec and prec_diag");
    if ((ds->prior_mu == NULL) != (ds->prior_prec == NULL)) return bf_set_error(BFHIP_ERR_ARG, "prior_mu and prior_prec go together");
    if ((pm.n2 > 0 && (!pm.mask2 || !pm.cubic2)) || (pm.n3 > 0 && (!pm.mask3 || !pm.cubic3)) || pm.n2 < 0 || pm.n3 < 0)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_pipeline_upload: cubic configs need their masks and coefficients");
    const bool all_linear = !pm.quad && pm.n2 == 0 && pm.n3 == 0;
    const bool use_bound = pm.use_bound && !all_linear;   // modules/poly.py:467
    if (use_bound && (!pm.mu || !pm.hess || !pm.f_mu || !(pm.alpha > 0.)))
        return bf_set_error(BFHIP_ERR_ARG, "use_bound needs mu, hess, f_mu and alpha > 0");

    // ---- transforms, bound and decay through the density upload (no polynomial of its own) ----
    bfhip_density_desc bd;
    memset(&bd, 0, sizeof(bd));
    bd.d = d;
    bd.ranges = ds->ranges; bd.hard_bounds = ds->hard_bounds; bd.su_lo = ds->su_lo; bd.su_diff = ds->su_diff;
    bd.use_bound = use_bound ? 1 : 0;
    bd.mu = pm.mu; bd.hess = pm.hess; bd.alpha = pm.alpha;
    bd.use_decay = ds->use_decay; bd.decay_mu = ds->decay_mu; bd.decay_hess = ds->decay_hess;
    bd.decay_alpha2 = ds->decay_alpha2; bd.decay_gamma = ds->decay_gamma;
    if (int rc = bfhip_density_upload(ctx, &bd)) return rc;
    ctx->has_model = 0;   // (until the pipeline part is in place)
    DevModel &dm = ctx->model;
    const int DP = dm.DP, ONE = DP, ZERO = DP + 1;

    // ---- monomials with a nonzero coefficient in some output, and their coefficient columns ----
    std::vector<Mono> mono;
    std::vector<std::vector<double>> col;   // col[p][o]
    auto add = [&](int a, int b, int c, auto coef_of) {
        std::vector<double> v(m);
        bool any = false;
        for (int o = 0; o < m; ++o) { v[o] = coef_of(o); any = any || v[o] != 0.; }
        if (!any) return;
        mono.push_back(Mono{{a, b, c}});
        col.push_back(std::move(v));
    };
    {   // the constant is always there (padding monomials are zeros, not ones)
        std::vector<double> v(m);
        for (int o = 0; o < m; ++o) v[o] = pm.c0[o];
        mono.push_back(Mono{{ONE, ONE, ONE}});
        col.push_back(std::move(v));
    }
    for (int j = 0; j < d; ++j) add(j, ONE, ONE, [&](int o) { return pm.lin[(size_t)o * d + j]; });
    if (pm.quad)
        for (int j = 0; j < d; ++j)
            for (int k = j; k < d; ++k) add(j, k, ONE, [&](int o) { return pm.quad[((size_t)o * d + j) * d + k]; });
    for (int a = 0; a < pm.n2; ++a)       // x_j^2 x_k, modules/_poly.pyx:49-84
        for (int b = 0; b < pm.n2; ++b) {
            const int j = pm.mask2[a], k = pm.mask2[b];
            if (j < 0 || j >= d || k < 0 || k >= d) return bf_set_error(BFHIP_ERR_ARG, "mask2 out of range");
            add(j, j, k, [&](int o) { return pm.cubic2[((size_t)o * pm.n2 + a) * pm.n2 + b]; });
        }
    for (int a = 0; a < pm.n3; ++a)       // x_j x_k x_l, j < k < l, :86-137
        for (int b = a + 1; b < pm.n3; ++b)
            for (int c = b + 1; c < pm.n3; ++c) {
                const int j = pm.mask3[a], k = pm.mask3[b], l = pm.mask3[c];
                if (j < 0 || j >= d || k < 0 || k >= d || l < 0 || l >= d) return bf_set_error(BFHIP_ERR_ARG, "mask3 out of range");
                add(j, k, l, [&](int o) { return pm.cubic3[(((size_t)o * pm.n3 + a) * pm.n3 + b) * pm.n3 + c]; });
            }
    const int nf = (int)mono.size(), PP = roundup(nf, 16);
    int MP = roundup(m, 16);
    if (PP >= 65536) return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_pipeline_upload: %d monomials", nf);

    // ---- whitening: prec = L L^T; C' = L^T C, y' = L^T y, f_mu' = L^T f_mu ----
    std::vector<double> Cw((size_t)MP * PP, 0.), yw(MP, 0.), fmuw(MP, 0.);
    if (ds->prec) {
        std::vector<double> L;
        if (!cholesky_lower(ds->prec, m, L))
            return bf_set_error(BFHIP_ERR_ARG, "bfhip_pipeline_upload: the precision matrix is not positive definite");
        for (int i = 0; i < m; ++i) {
            double sy = 0., sf = 0.;
            for (int k = i; k < m; ++k) {
                const double l = L[(size_t)k * m + i];
                sy += l * ds->y[k];
                if (use_bound) sf += l * pm.f_mu[k];
            }
            yw[i] = sy;
            fmuw[i] = sf;
        }
        for (int p = 0; p < nf; ++p) {
            const double *cp = col[p].data();
            for (int i = 0; i < m; ++i) {
                double s = 0.;
                for (int k = i; k < m; ++k) s += L[(size_t)k * m + i] * cp[k];
                Cw[(size_t)i * PP + p] = s;
            }
        }
    } else {
        for (int i = 0; i < m; ++i) {
            const double pd = ds->prec_diag[i];
            if (!(pd >= 0.) || !std::isfinite(pd)) return bf_set_error(BFHIP_ERR_ARG, "bfhip_pipeline_upload: prec_diag should be non-negative");
            const double s = std::sqrt(pd);
            yw[i] = s * ds->y[i];
            fmuw[i] = use_bound ? s * pm.f_mu[i] : 0.;
            for (int p = 0; p < nf; ++p) Cw[(size_t)i * PP + p] = s * col[p][i];
        }
    }

    // ---- output-space compression: more outputs than monomials -> m_eff = nf, exactly ----
    // f_0 = C' phi lives in the column space of C' (at most nf dimensions of the m): with the Householder factorisation
    // C' = Q [R; 0] (Q m x m orthogonal), |a f_0 - b f_mu' - y'|^2 = |a R phi - b (Q^T f_mu')_head - (Q^T y')_head|^2 + |b (Q^T f_mu')_tail
    // + (Q^T y')_tail|^2, and J_0^T r = (d phi)^T R^T r_head: the device works with R (nf x nf), the heads of Q^T y' and Q^T f_mu',
    // and three scalars of the tails (|y_tail|^2 goes into logp0; |f_mu_tail|^2 and f_mu_tail . y_tail enter only outside the
    // bound, where b = (beta - alpha) / alpha is not zero).  At the DES shape (m = 457, nf = 73) both contractions shrink 6.3x.
    double k_ff = 0., k_fy = 0., k_yy = 0.;
    int m_eff = m;
    const bool compress = !bf_tune().pld_no_compress && m > nf;
    if (compress) {
        // Householder QR of the m x nf block of Cw, in place; the reflectors are applied to yw and fmuw as they are formed
        std::vector<double> v(m);
        for (int j = 0; j < nf; ++j) {
            double nrm = 0.;
            for (int i = j; i < m; ++i) nrm += Cw[(size_t)i * PP + j] * Cw[(size_t)i * PP + j];
            nrm = std::sqrt(nrm);
            if (!(nrm > 0.)) continue;   // (a zero column below the diagonal: nothing to reflect)
            const double x0 = Cw[(size_t)j * PP + j];
            const double alpha_h = x0 > 0. ? -nrm : nrm;
            double vnorm2 = 0.;
            for (int i = j; i < m; ++i) { v[i] = Cw[(size_t)i * PP + j]; }
            v[j] -= alpha_h;
            for (int i = j; i < m; ++i) vnorm2 += v[i] * v[i];
            if (!(vnorm2 > 0.)) continue;
            const double tau = 2. / vnorm2;
            auto reflect_col = [&](auto get, auto put) {
                double dotv = 0.;
                for (int i = j; i < m; ++i) dotv += v[i] * get(i);
                const double sc = tau * dotv;
                for (int i = j; i < m; ++i) put(i, get(i) - sc * v[i]);
            };
            for (int c = j; c < nf; ++c)
                reflect_col([&](int i) { return Cw[(size_t)i * PP + c]; }, [&](int i, double val) { Cw[(size_t)i * PP + c] = val; });
            reflect_col([&](int i) { return yw[i]; }, [&](int i, double val) { yw[i] = val; });
            reflect_col([&](int i) { return fmuw[i]; }, [&](int i, double val) { fmuw[i] = val; });
            for (int i = j + 1; i < m; ++i) Cw[(size_t)i * PP + j] = 0.;   // (exactly: below the diagonal of R)
        }
        for (int i = nf; i < m; ++i) {
            k_ff += fmuw[i] * fmuw[i];
            k_fy += fmuw[i] * yw[i];
            k_yy += yw[i] * yw[i];
        }
        m_eff = nf;
        MP = roundup(m_eff, 16);
        std::vector<double> C2((size_t)MP * PP, 0.), y2(MP, 0.), f2(MP, 0.);
        for (int i = 0; i < m_eff; ++i) {
            for (int c = 0; c < nf; ++c) C2[(size_t)i * PP + c] = Cw[(size_t)i * PP + c];
            y2[i] = yw[i];
            f2[i] = fmuw[i];
        }
        Cw.swap(C2); yw.swap(y2); fmuw.swap(f2);
    }
    const int NT1 = MP / 16, NS1 = PP / 4, NT2 = PP / 16, NS2 = MP / 4;

    // ---- A fragments of C' and C'^T ----
    std::vector<double> CF((size_t)NT1 * NS1 * 64), CTF((size_t)NT2 * NS2 * 64);
    for (int t = 0; t < NT1; ++t)
        for (int s = 0; s < NS1; ++s)
            for (int l = 0; l < 64; ++l) CF[((size_t)t * NS1 + s) * 64 + l] = Cw[(size_t)(16 * t + (l & 15)) * PP + 4 * s + (l >> 4)];
    for (int u = 0; u < NT2; ++u)
        for (int s = 0; s < NS2; ++s)
            for (int l = 0; l < 64; ++l) CTF[((size_t)u * NS2 + s) * 64 + l] = Cw[(size_t)(4 * s + (l >> 4)) * PP + 16 * u + (l & 15)];

    // ---- monomial table and, per dimension, the monomials that contain it with their cofactors ----
    std::vector<unsigned> mono_tab(PP, (unsigned)ZERO | ((unsigned)ZERO << 8) | ((unsigned)ZERO << 16));
    std::vector<std::vector<unsigned long long>> per_dim(DP);
    for (int p = 0; p < nf; ++p) {
        const int *ix = mono[p].i;
        mono_tab[p] = (unsigned)ix[0] | ((unsigned)ix[1] << 8) | ((unsigned)ix[2] << 16);
        for (int j = 0; j < d; ++j) {
            int e = 0;
            for (int q = 0; q < 3; ++q) e += ix[q] == j;
            if (!e) continue;
            // d/dx_j of x_j^e * rest = e x_j^(e-1) rest: the two remaining factors (ones where the monomial is shorter)
            int rest[3], nr = 0;
            bool dropped = false;
            for (int q = 0; q < 3; ++q) {
                if (ix[q] == j && !dropped) { dropped = true; continue; }
                rest[nr++] = ix[q];
            }
            const unsigned hi = (unsigned)rest[0] | ((unsigned)rest[1] << 8) | ((unsigned)e << 16);
            per_dim[j].push_back((unsigned long long)(unsigned)p | ((unsigned long long)hi << 32));
        }
    }
    size_t n_ent = 0;
    for (int j = 0; j < DP; ++j) n_ent = per_dim[j].size() > n_ent ? per_dim[j].size() : n_ent;
    const unsigned long long pad_ent = (unsigned long long)0u | ((unsigned long long)((unsigned)ZERO | ((unsigned)ZERO << 8) | (1u << 16)) << 32);
    std::vector<unsigned long long> gtab(n_ent * DP, pad_ent);
    for (int j = 0; j < DP; ++j)
        for (size_t i = 0; i < per_dim[j].size(); ++i) gtab[i * DP + j] = per_dim[j][i];

    // ---- K-split of GEMM2: the fewest rounds x steps per job that fits the CU's LDS with the sampler's own regions ----
    dm.pld.on = 1;   // (the LDS size below depends on it)
    const size_t base_bytes = bf_sampler_lds_bytes_base(dm);
    int best_ks = 0, only8 = 0;
    long best_cost = 0;
    // (the sixteen-chain forms' layout first -- every form can run then --, else the eight-chain forms' compact rows)
    for (int xs : {PLD_XS, PLD_XS8}) {
        for (int ks = 1; ks <= PLD_MAX_KS2; ++ks) {
            const int kpj = roundup((NS2 + ks - 1) / ks, 4);
            if (ks > 1 && (ks - 1) * kpj >= NS2) continue;   // an empty part
            if (base_bytes + pld_lds_doubles(DP, MP, PP, ks, (int)n_ent, xs) * sizeof(double) > (size_t)160 * 1024) continue;
            const long cost = (long)((NT2 * ks + 15) / 16) * kpj;
            if (!best_ks || cost < best_cost) { best_ks = ks; best_cost = cost; }
        }
        if (best_ks) { only8 = xs == PLD_XS8; break; }
    }
    if (!best_ks) {
        dm.pld.on = 0;
        return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_pipeline_upload: %d outputs x %d monomials need %zu KB of LDS per workgroup (160 KB)",
                            m, nf, (base_bytes + pld_lds_doubles(DP, MP, PP, 1, (int)n_ent, PLD_XS8) * sizeof(double)) / 1024);
    }

    // ---- one device buffer: doubles, then 8-byte entries, then the monomial words ----
    std::vector<double> prior(2 * (size_t)DP, 0.);
    if (ds->prior_mu)
        for (int i = 0; i < d; ++i) {
            prior[i] = ds->prior_mu[i];
            prior[DP + i] = ds->prior_prec[i];
            if (!(ds->prior_prec[i] >= 0.)) return bf_set_error(BFHIP_ERR_ARG, "bfhip_pipeline_upload: prior_prec should be non-negative");
        }
    const size_t n_dbl = CF.size() + CTF.size() + 2 * (size_t)MP + prior.size();
    const size_t bytes = n_dbl * 8 + gtab.size() * 8 + mono_tab.size() * 4 + 64;
    if (ctx->pld_bytes < bytes) {
        if (ctx->pld_buf) BF_HIP_CHECK(hipFree(ctx->pld_buf));
        ctx->pld_buf = NULL;
        ctx->pld_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->pld_buf, bytes));
        ctx->pld_bytes = bytes;
    }
    std::vector<char> hb(bytes, 0);
    double *hd = (double *)hb.data();
    size_t o = 0;
    const size_t o_cf = o; memcpy(hd + o, CF.data(), CF.size() * 8); o += CF.size();
    const size_t o_ctf = o; memcpy(hd + o, CTF.data(), CTF.size() * 8); o += CTF.size();
    const size_t o_y = o; memcpy(hd + o, yw.data(), (size_t)MP * 8); o += MP;
    const size_t o_f = o; memcpy(hd + o, fmuw.data(), (size_t)MP * 8); o += MP;
    const size_t o_pr = o; memcpy(hd + o, prior.data(), prior.size() * 8); o += prior.size();
    const size_t o_g = o; memcpy(hd + o, gtab.data(), gtab.size() * 8); o += gtab.size();
    const size_t o_m = o; memcpy(hd + o, mono_tab.data(), mono_tab.size() * 4);
    BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    BF_HIP_CHECK(hipMemcpy(ctx->pld_buf, hb.data(), bytes, hipMemcpyHostToDevice));
    const double *dbase = (const double *)ctx->pld_buf;
    PldDev &pl = dm.pld;
    pl.on = 1;
    pl.m = m_eff; pl.m_full = m; pl.MP = MP; pl.NT1 = NT1; pl.NS2 = NS2;
    pl.k_ff = k_ff; pl.k_fy = k_fy;
    pl.nf = nf; pl.PP = PP; pl.NS1 = NS1; pl.NT2 = NT2;
    pl.KS2 = best_ks; pl.KPJ2 = roundup((NS2 + best_ks - 1) / best_ks, 4);
    pl.n_ent = (int)n_ent;
    pl.only8 = only8;
    pl.has_prior = ds->prior_mu != NULL;
    pl.tri = compress ? 1 : 0;   // (the Householder factorisation left R: zero below the diagonal, exactly)
    pl.CF = dbase + o_cf; pl.CTF = dbase + o_ctf; pl.yw = dbase + o_y; pl.fmuw = dbase + o_f;
    pl.prior_mu = dbase + o_pr; pl.prior_prec = dbase + o_pr + DP;
    pl.gtab = (const unsigned long long *)(dbase + o_g);
    pl.mono = (const unsigned *)(dbase + o_m);
    pl.logp0 = ds->logp0 - 0.5 * k_yy;   // (the part of the data vector outside the surrogate's column space: a constant)
    pl.prior_c0 = ds->prior_c0;
    ctx->has_model = 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Density.logp_and_grad (core/density.py:724-754) of the pipeline density for n points: one workgroup of 16 waves per 16
// points, wave = point, lane = dimension; the two contractions are shared by the workgroup as in the sampler.
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline double pld_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// element (i, k) of a DP x DP matrix stored as A fragments
__device__ inline double pld_frag_at(const double *Mf, int DP, int i, int k) {
    return Mf[(((i >> 4) * (DP / 4)) + (k >> 2)) * 64 + (i & 15) + 16 * (k & 3)];
}

// NPT points per workgroup: 16 (sixteen waves, 16-column tiles) or 8 (eight waves, the eight-chain forms' compact LDS rows: what
// a surrogate with more monomials than the sixteen-point layout holds runs on); E dimensions per lane (2 at d > 64)
template <int NPT, int E>
__global__ __launch_bounds__(NPT * 64) void bf_pld_logp_grad_kernel(DevModel m, int n, const double *__restrict__ x, int original_space,
                                                                  double *__restrict__ logp, double *__restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const PldDev &pl = m.pld;
    const int DP = m.DP, d = m.d;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const PldLds L = pld_lds(lds, DP, pl, NPT);
    double *XM = lds + pld_lds_doubles(DP, pl.MP, pl.PP, pl.KS2, pl.n_ent, L.XS);   // [NPT][2][DP]  x - mu and x_o - mu_decay of every point
    pld_stage(pl, L, DP, tid, NPT * 64);
    const bool tr = m.has_transform && !original_space;
    for (int base = blockIdx.x * NPT; base < n; base += gridDim.x * NPT) {
        const int i = base + w;
        const bool valid = i < n;
        double xo[E], jac[E], gj[E], su_diff[E], xs[E], xm[E], xd[E], hv[E], dgr[E], mu[E];
        double logdet = 0.;
        bool on[E];
        double *xmw = XM + (size_t)w * 2 * DP;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = lane * E + e;
            on[e] = dim < d;
            const double xin = (valid && on[e]) ? x[(size_t)i * d + dim] : 0.;
            xo[e] = xin;
            jac[e] = 1.;
            gj[e] = 0.;
            if (tr && on[e]) {
                double J, J2;
                bf_to_original(xin, (int)m.pd[PD_KIND * DP + dim], m.pd[PD_LO * DP + dim], m.pd[PD_RG * DP + dim], xo[e], J, J2);
                logdet += log(fabs(J));
                jac[e] = J;
                gj[e] = J2 / J;
            }
            su_diff[e] = (m.has_su && on[e]) ? m.pd[PD_SU_DIFF * DP + dim] : 1.;
            xs[e] = (m.has_su && on[e]) ? (xo[e] - m.pd[PD_SU_LO * DP + dim]) / su_diff[e] : xo[e];
            mu[e] = on[e] ? m.pd[PD_MU * DP + dim] : 0.;
            xm[e] = on[e] ? xs[e] - mu[e] : 0.;
            xd[e] = (m.use_decay && on[e]) ? xo[e] - m.pd[PD_DMU * DP + dim] : 0.;
            if (dim < DP) { xmw[dim] = xm[e]; xmw[DP + dim] = xd[e]; }
        }
        double b2 = 0., bd2 = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = lane * E + e;
            hv[e] = 0.;
            dgr[e] = 0.;
            if (dim < DP) {
                if (m.use_bound)
                    for (int k = 0; k < d; ++k) hv[e] += pld_frag_at(m.Hf, DP, dim, k) * xmw[k];
                if (m.use_decay)
                    for (int k = 0; k < d; ++k) dgr[e] += pld_frag_at(m.Hdf, DP, dim, k) * xmw[DP + k];
            }
            b2 += xm[e] * hv[e];
            bd2 += xd[e] * dgr[e];
        }
        const double r_b2 = pld_wave_sum(b2), r_bd2 = pld_wave_sum(bd2);
        logdet = pld_wave_sum(logdet);
        double beta = 0.;
        if (m.use_bound) {   // modules/poly.py:467-469
            const double b = sqrt(r_b2);
            if (b > m.alpha) beta = b;
        }
        double x_eval[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double xv = beta > 0. ? (m.alpha * xs[e] + (beta - m.alpha) * mu[e]) / beta : xs[e];   // :482
            x_eval[e] = (valid && on[e]) ? xv : 0.;
        }
        pld_point_e<E>(pl, L, DP, w, lane, x_eval, valid ? beta : 0.);
        __syncthreads();
        if constexpr (NPT == 8) pld_gemm1_q8(pl, L, m.alpha, w, 8, lane);
        else pld_gemm1(pl, L, m.alpha, w, 16, lane);
        __syncthreads();
        if constexpr (NPT == 8) pld_gemm2_q8(pl, L, w, 8, lane);
        else pld_gemm2(pl, L, w, 16, lane);
        __syncthreads();
        double s_rr, s_fr;
        pld_sums(pl, L, w, lane, NPT, s_rr, s_fr);
        s_rr = pld_wave_sum(s_rr);
        s_fr = pld_wave_sum(s_fr);
        double gn[E], dj = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            gn[e] = lane * E + e < DP ? pld_grad(pl, L, DP, w, lane * E + e) : 0.;   // (J_0^T r)_dim
            dj += gn[e] * xm[e];
        }
        if (beta > 0.) {   // (compressed outputs: the tails of Q^T f_mu' and Q^T y' as scalars, bfhip_pipeline_upload)
            const double b = (beta - m.alpha) / m.alpha;
            s_rr += b * (b * pl.k_ff + 2. * pl.k_fy);
            s_fr += b * pl.k_ff + pl.k_fy;
            const double r_dotj = pld_wave_sum(dj);   // modules/poly.py:494-496 contracted with r
#pragma unroll
            for (int e = 0; e < E; ++e) gn[e] += (s_fr / m.alpha - r_dotj / beta) * (hv[e] / beta);
        }
        double f = pl.logp0 - 0.5 * s_rr;
        double g[E], pr = 0.;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dim = lane * E + e;
            g[e] = -gn[e];
            if (m.has_su) g[e] = g[e] / su_diff[e];   // core/module.py:226
            g[e] = g[e] * jac[e];                      // density.py:558
            if (pl.has_prior) {
                const double dx = on[e] ? xo[e] - pl.prior_mu[dim] : 0., pp = on[e] ? pl.prior_prec[dim] : 0.;
                pr += pp * dx * dx;
                g[e] += -(pp * dx) * jac[e];
            }
        }
        if (pl.has_prior) f += pl.prior_c0 - 0.5 * pld_wave_sum(pr);
        if (m.use_decay) {   // density.py:740-746
            f -= m.decay_gamma * bf_clip0(r_bd2 - m.decay_alpha2);
            if (r_bd2 > m.decay_alpha2) {
#pragma unroll
                for (int e = 0; e < E; ++e) g[e] -= 2. * m.decay_gamma * dgr[e];
            }
        }
        if (tr) {            // :747-750
            f += logdet;
#pragma unroll
            for (int e = 0; e < E; ++e) g[e] += gj[e];
        }
        if (valid) {
            if (lane == 0) logp[i] = f;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (grad && on[e]) grad[(size_t)i * d + lane * E + e] = g[e];
        }
        __syncthreads();   // the LDS regions are rewritten by the next batch
    }
}

int bf_pld_logp_grad(bfhip_ctx *ctx, int n, const double *x, int original_space, double *logp, double *grad) {
    const DevModel &m = ctx->model;
    const int npt = m.pld.only8 ? 8 : 16;
    const size_t lds = (pld_lds_doubles(m.DP, m.pld.MP, m.pld.PP, m.pld.KS2, m.pld.n_ent, npt == 8 ? PLD_XS8 : PLD_XS) + (size_t)npt * 2 * m.DP) * sizeof(double);
    if (lds > (size_t)160 * 1024) return bf_set_error(BFHIP_ERR_UNSUPPORTED, "pipeline density: %zu KB of LDS", lds / 1024);
    const bool e2 = m.DP > 64;   // two dimensions per lane
    void (*k)(DevModel, int, const double *, int, double *, double *) =
        npt == 8 ? (e2 ? bf_pld_logp_grad_kernel<8, 2> : bf_pld_logp_grad_kernel<8, 1>) : (e2 ? bf_pld_logp_grad_kernel<16, 2> : bf_pld_logp_grad_kernel<16, 1>);
    if (lds > 64 * 1024) BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int grid = (n + npt - 1) / npt;
    if (grid > 4 * ctx->n_cu) grid = 4 * ctx->n_cu;
    hipLaunchKernelGGL(k, dim3(grid), dim3(npt * 64), lds, ctx->stream, m, n, x, original_space, logp, grad);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
