// bfhip_tnuts.hip -- tempered NUTS: BaseTHMC.astep (samplers/hmc_utils/base_hmc.py:220-262) around
// the NUTS tree (samplers/nuts.py:21-217, TTree: samplers/tnuts.py:15-41) with TCpuLeapfrogIntegrator
// (samplers/hmc_utils/integration.py:98-222): the state carries a tempering coordinate u with momentum v, the potential
// is beta(u) phi + (1 - beta(u)) psi + U(u) with phi = -logp of the surrogate target and psi = -(logp of a base density
// + log xi), and every leapfrog step evaluates both densities twice (mid-point gradients, end-point values).
//
// Layout: ONE WAVE PER CHAIN, lane = dimension (d <= 64), no exchange between chains: a coverage-and-parity kernel, not
// the throughput path (the target's and the base's matrices are staged once per workgroup in LDS, transposed, and every
// matvec is 64 broadcast-FMA steps per lane).  Target: the common surrogate (linear + quadratic configs with the
// extrapolation bound, no transform / scaling / decay / cubic).  Base: a quadratic log-density without bound (e.g. the
// Gaussian approximation of the posterior).  Draws are consumed in the recursion's post-order, as in the other sampler
// kernels, so a chain reproduces the CPU oracle for the same xoshiro stream.
#include <cmath>
#include "bfhip_common.h"
#include "bfhip_sampler_defs.h"

#define TN_MAXL BFHIP_MAX_TREEDEPTH
enum { TS_LS = 0, TS_ACC, TS_E, TS_LOGP, TS_U, TS_W, TS_N };  // per-level stack scalars

struct TnutsArgs {
    bfhip_sampler_config cfg;
    int n_chain, iter_end, iter_out0, n_out, d;
    uint64_t *rng;
    double *sc, *vec, *tu, *samples, *stats, *stats_t;
    unsigned long long *n_leapfrog;
    double *scratch;  // [n_chain][4 * TN_MAXL][64] subtree stack vectors
    const double *base_S, *base_lin;  // (d,d) symmetric S_b = A_b + A_b^T, (d,)
    double base_c0, logxi;
};

__device__ inline double tn_wsum(double v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
    return v;
}
__device__ inline double tn_logaddexp(double a, double b) {
    const double mx = a > b ? a : b, mn = a > b ? b : a;
    return (mx == -INFINITY) ? -INFINITY : mx + log1p(exp(mn - mx));
}

template <int WPB>
__global__ __launch_bounds__(64 * WPB) void bf_tnuts_kernel(DevModel m, TnutsArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int d = a.d;
    double *St = lds;                 // [64][64] transposed: St[k * 64 + row]
    double *Ht = St + 4096;
    double *Bt = Ht + 4096;
    double *XS = Bt + 4096;           // [WPB][64] broadcast buffer of the wave
    double *LSC = XS + WPB * 64;      // [WPB][TN_MAXL][TS_N]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int chain = blockIdx.x * WPB + w;
    const bool real = chain < a.n_chain;
    const int NS = m.DP / 4;
    for (int i = threadIdx.x; i < 4096; i += 64 * WPB) {
        const int k = i >> 6, row = i & 63;
        double s = 0., h = 0., b = 0.;
        if (row < d && k < d) {
            const size_t fi = ((size_t)(row / 16) * NS + k / 4) * 64 + (row % 16) + 16 * (k % 4);  // A fragments -> M[row][k]
            s = m.Sf[fi];
            h = m.Hf[fi];
            b = a.base_S[(size_t)row * d + k];
        }
        St[i] = s; Ht[i] = h; Bt[i] = b;
    }
    __syncthreads();
    if (!real) return;  // (no barrier below: every wave is on its own)
    double *lsw = LSC + w * (TN_MAXL * TS_N);
    const bool in = lane < d;
    const double c_lin = in ? m.pd[PD_LIN * m.DP + lane] : 0., c_mu = in ? m.pd[PD_MU * m.DP + lane] : 0.;
    const double b_lin = in ? a.base_lin[lane] : 0.;
    // x_k of the wave's vector as a scalar broadcast (no LDS round trip)
    auto bcast = [&](double x, int k) -> double {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k));
    };
    // M x with the matrix in LDS by columns: four accumulation chains (k mod 4), added at the end.  (The rows' sums associate
    // differently from a single chain over k: rounding-level, inside the tolerance of the oracle comparison.)
    auto matvec = [&](const double *Mt, double x) -> double {
        double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
        int k = 0;
        for (; k + 4 <= d; k += 4) {
            const double m0 = Mt[k * 64 + lane], m1 = Mt[(k + 1) * 64 + lane], m2 = Mt[(k + 2) * 64 + lane], m3 = Mt[(k + 3) * 64 + lane];
            a0 = __builtin_fma(m0, bcast(x, k), a0);
            a1 = __builtin_fma(m1, bcast(x, k + 1), a1);
            a2 = __builtin_fma(m2, bcast(x, k + 2), a2);
            a3 = __builtin_fma(m3, bcast(x, k + 3), a3);
        }
        for (; k < d; ++k) a0 = __builtin_fma(Mt[k * 64 + lane], bcast(x, k), a0);
        return (a0 + a1) + (a2 + a3);
    };
    // the three products every evaluation needs -- S q, H (q - mu), S_b q -- in ONE pass over k: six independent chains, the
    // broadcasts of q shared by S and S_b
    auto matvec3 = [&](double q, double xm, double &sx, double &hv, double &bx) {
        double s0 = 0., s1 = 0., h0 = 0., h1 = 0., b0 = 0., b1 = 0.;
        int k = 0;
        for (; k + 2 <= d; k += 2) {
            const double ms0 = St[k * 64 + lane], mh0 = Ht[k * 64 + lane], mb0 = Bt[k * 64 + lane];
            const double ms1 = St[(k + 1) * 64 + lane], mh1 = Ht[(k + 1) * 64 + lane], mb1 = Bt[(k + 1) * 64 + lane];
            const double q0 = bcast(q, k), q1 = bcast(q, k + 1), x0 = bcast(xm, k), x1 = bcast(xm, k + 1);
            s0 = __builtin_fma(ms0, q0, s0); h0 = __builtin_fma(mh0, x0, h0); b0 = __builtin_fma(mb0, q0, b0);
            s1 = __builtin_fma(ms1, q1, s1); h1 = __builtin_fma(mh1, x1, h1); b1 = __builtin_fma(mb1, q1, b1);
        }
        if (k < d) {
            const double q0 = bcast(q, k), x0 = bcast(xm, k);
            s0 = __builtin_fma(St[k * 64 + lane], q0, s0); h0 = __builtin_fma(Ht[k * 64 + lane], x0, h0); b0 = __builtin_fma(Bt[k * 64 + lane], q0, b0);
        }
        sx = s0 + s1; hv = h0 + h1; bx = b0 + b1;
    };
    // phi, dphi, psi, dpsi at q (this lane's coordinate): integration.py:180-181 / base_hmc.py:227-231
    auto potentials = [&](double q, double &phi, double &dphi, double &psi, double &dpsi) {
        // target surrogate with its bound (modules/poly.py:466-503)
        const double xm = in ? q - c_mu : 0.;
        double sx, hv, bx;
        matvec3(q, xm, sx, hv, bx);
        double gn = sx + c_lin;
        double f = m.c0 + tn_wsum(in ? __builtin_fma(0.5 * q, sx, c_lin * q) : 0.);
        const double beta = sqrt(tn_wsum(xm * hv));
        if (beta > m.alpha) {
            const double x0 = in ? (m.alpha * q + (beta - m.alpha) * c_mu) / beta : 0.;
            sx = matvec(St, x0);
            const double j0 = sx + c_lin;
            const double f0 = m.c0 + tn_wsum(in ? __builtin_fma(0.5 * x0, sx, c_lin * x0) : 0.);
            const double dotj = tn_wsum(in ? j0 * xm : 0.);
            f = (beta * f0 - (beta - m.alpha) * m.f_mu) / m.alpha;
            gn = j0 + ((f0 - m.f_mu) / m.alpha - dotj / beta) * (hv / beta);
        }
        phi = -f;
        dphi = in ? -gn : 0.;
        // base: c0 + lin.x + x.S_b x / 2, plus log xi
        const double fb = a.base_c0 + tn_wsum(in ? __builtin_fma(0.5 * q, bx, b_lin * q) : 0.);
        psi = -(fb + a.logxi);
        dpsi = in ? -(bx + b_lin) : 0.;
    };
    auto t_beta = [](double u) { return 1 / (1 + exp(-u)); };
    auto t_pot = [](double u) { return u + 2 * log(1 + exp(-u)); };

    // ---- chain state ----
    double *scp = a.sc + (size_t)chain * BFHIP_SC_N;
    double *vecp = a.vec + (size_t)chain * BFHIP_VEC_N * d;
    double *sb = a.scratch + (size_t)chain * (4 * TN_MAXL) * 64 + lane;
    uint64_t rs[4];
    for (int k = 0; k < 4; ++k) rs[k] = a.rng[(size_t)chain * 4 + k];
    double log_step = scp[BFHIP_SC_LOG_STEP], log_bar = scp[BFHIP_SC_LOG_BAR], hbar = scp[BFHIP_SC_HBAR];
    const double smu = scp[BFHIP_SC_MU];
    double count = scp[BFHIP_SC_COUNT];
    double fg_n = scp[BFHIP_SC_FG_N], bg_n = scp[BFHIP_SC_BG_N], n_samples = scp[BFHIP_SC_N_SAMPLES];
    double prev_upd = scp[BFHIP_SC_PREV_UPDATE], adapt_window = scp[BFHIP_SC_ADAPT_WINDOW];
    int i_iter = (int)scp[BFHIP_SC_I_ITER], err = (int)scp[BFHIP_SC_ERROR];
    double qc = in ? vecp[BFHIP_VEC_Q * d + lane] : 0., var = in ? vecp[BFHIP_VEC_VAR * d + lane] : 1.;
    double u_cur = a.tu[chain];
    unsigned long long nlf = 0;
    auto uni = [&]() { return bf_u01(bf_xoshiro_next(rs)); };
    auto logbern = [&](double l) -> bool {  // nuts.py:200-203
        if (l != l) err = 2;
        return log(uni()) < l;
    };

    // one tempered leapfrog step from (q, p, u, vt): integration.py:153-222
    struct TS { double q, p, u, vt, weight, energy, logp; };
    auto finish_state = [&](TS &s, double phi, double psi) {
        const double kin = tn_wsum(s.p * (var * s.p));
        const double beta = t_beta(s.u);
        s.energy = (beta * phi + (1 - beta) * psi + t_pot(s.u)) + (0.5 * kin + s.vt * s.vt / 2);
        s.logp = -phi;
        const double delta = phi - psi;
        s.weight = (delta == 0) ? 1. : delta / expm1(delta);
    };
    auto t_step = [&](const TS &s0, double eps) -> TS {
        TS s = s0;
        const double dt = 0.5 * eps;
        double phi, dphi, psi, dpsi;
        s.u += s.vt * dt;
        s.q += dt * (var * s.p);
        potentials(s.q, phi, dphi, psi, dpsi);
        const double beta = t_beta(s.u);
        const double e = exp(-s.u), dbeta = e / ((1 + e) * (1 + e));
        const double eu = exp(s.u), dU = (eu - 1) / (eu + 1);
        s.vt += -(dbeta * (phi - psi) + dU) * eps;
        s.p += eps * -(beta * dphi + (1 - beta) * dpsi);
        s.u += s.vt * dt;
        s.q += dt * (var * s.p);
        potentials(s.q, phi, dphi, psi, dpsi);
        finish_state(s, phi, psi);
        return s;
    };

    while (i_iter < a.iter_end && err == 0) {
        const bool warm = i_iter < a.cfg.n_warmup;
        // ---- BaseTHMC.astep: base_hmc.py:233-262 ----
        TS start;
        start.q = qc;
        {   // p0 = metric.random: one xoshiro draw keys the SplitMix64 stream of the d normals (as in the other kernels)
            const uint64_t K = bf_xoshiro_next(rs);
            const uint64_t P = (uint64_t)(lane >> 1);
            const double u1 = bf_u01_open0(bf_mix64(K + (2 * P + 1) * BF_GOLDEN)), u2 = bf_u01(bf_mix64(K + (2 * P + 2) * BF_GOLDEN));
            const double rad = sqrt(-2. * log(u1));
            double sn, cs;
            sincospi(2. * u2, &sn, &cs);
            start.p = in ? (1. / sqrt(var)) * ((lane & 1) ? rad * sn : rad * cs) : 0.;
        }
        {   // v0 = rng.normal(0, 1): a stream of its own, first (cosine) element
            const uint64_t K = bf_xoshiro_next(rs);
            const double u1 = bf_u01_open0(bf_mix64(K + BF_GOLDEN)), u2 = bf_u01(bf_mix64(K + 2 * BF_GOLDEN));
            double sn, cs;
            sincospi(2. * u2, &sn, &cs);
            start.vt = sqrt(-2. * log(u1)) * cs;
        }
        start.u = u_cur;
        {
            double phi, dphi, psi, dpsi;
            potentials(start.q, phi, dphi, psi, dpsi);
            finish_state(start, phi, psi);
        }
        if (!(fabs(start.energy) <= 1.7976931348623157e308)) { err = 1; break; }
        const double eps0 = exp(warm ? log_step : log_bar);
        // ---- Tree.__init__: nuts.py:24-43 ----
        TS left = start, right = start;
        double prop_q = start.q, prop_u = start.u, prop_w = start.weight, prop_E = start.energy, prop_logp = start.logp;
        double p_sum = start.p, log_size = 0., accept_sum = 0., max_de = 0.;
        int depth = 0, n_prop = 0, diverging = 0, turning = 0;
        for (int it = 0; it < a.cfg.max_treedepth && err == 0; ++it) {
            const int dir = logbern(-0.6931471805599453094) ? 1 : -1;  // nuts.py:210
            const double eps = dir > 0 ? eps0 : -eps0;
            const TS old_left = left, old_right = right;
            // ---- _build_subtree(edge, depth, eps), recursion flattened: leaf i merges upwards while bit `lev` of i is set ----
            TS cur = dir > 0 ? right : left;
            // the subtree under construction: first state (T_l*), last state = cur, p_sum, proposal, log size, accept sum
            double T_lp = 0., T_ps = 0., T_pq = 0., T_pu = 0., T_pw = 0., T_pE = 0., T_plogp = 0., T_ls = 0., T_acc = 0.;
            // level 0 of the stack in registers
            double L0_lp = 0., L0_rp = 0., L0_ps = 0., L0_pq = 0.;
            double sub_acc = 0.;
            long sub_n = 0;
            bool done = false;
            const int n_leaf = 1 << depth;
            for (int i_leaf = 0; i_leaf < n_leaf && !done; ++i_leaf) {
                // ---- _single_step: nuts.py:105-132 ----
                const TS nxt = t_step(cur, eps);
                nlf += 1;
                sub_n += 1;
                double dE = nxt.energy - start.energy;
                if (dE != dE) dE = INFINITY;
                if (fabs(dE) > fabs(max_de)) max_de = dE;
                if (!(fabs(dE) < a.cfg.max_change)) {
                    diverging = 1;
                    // the stub subtree: ancestors still add their left halves' accept sums (nuts.py:173)
                    for (int al = 0; al < depth; ++al)
                        if ((i_leaf >> al) & 1) sub_acc += (al == 0 ? lsw[TS_ACC] : lsw[al * TS_N + TS_ACC]);
                    done = true;
                    break;
                }
                cur = nxt;
                T_lp = nxt.p; T_ps = nxt.p;
                T_pq = nxt.q; T_pu = nxt.u; T_pw = nxt.weight; T_pE = nxt.energy; T_plogp = nxt.logp;
                T_ls = -dE;
                { const double pa = exp(-dE); T_acc = pa > 1. ? 1. : pa; }
                int lev = 0;
                bool abort = false;
                while (lev < depth && ((i_leaf >> lev) & 1)) {
                    // ---- merge with the waiting left sibling of this level: nuts.py:146-178 ----
                    double A_lp, A_rp, A_ps, A_pq;  // sibling: left p, right p, p_sum, proposal q
                    double A_lv, A_rv;              // velocities of its ends
                    if (lev == 0) {
                        A_lp = L0_lp; A_rp = L0_rp; A_ps = L0_ps; A_pq = L0_pq;
                    } else {
                        A_lp = sb[(size_t)(4 * lev + 0) * 64]; A_rp = sb[(size_t)(4 * lev + 1) * 64];
                        A_ps = sb[(size_t)(4 * lev + 2) * 64]; A_pq = sb[(size_t)(4 * lev + 3) * 64];
                    }
                    A_lv = var * A_lp; A_rv = var * A_rp;
                    const double *ls = lsw + lev * TS_N;
                    const double psum = A_ps + T_ps;
                    bool turn = (tn_wsum(psum * A_lv) <= 0.) || (tn_wsum(psum * (var * cur.p)) <= 0.);
                    if (lev >= 1) {  // sub-span checks for depth > 1 (nuts.py:154-161)
                        const double ps1 = A_ps + T_lp;
                        const double ps2 = A_rp + T_ps;
                        turn = turn || (tn_wsum(ps1 * A_lv) <= 0.) || (tn_wsum(ps1 * (var * T_lp)) <= 0.);
                        turn = turn || (tn_wsum(ps2 * A_rv) <= 0.) || (tn_wsum(ps2 * (var * cur.p)) <= 0.);
                    }
                    const double acc_l = ls[TS_ACC], ls_l = ls[TS_LS];
                    const double ls_new = tn_logaddexp(ls_l, T_ls);
                    const bool take2 = logbern(T_ls - ls_new);  // :164 (drawn even when this merge turns)
                    T_acc = acc_l + T_acc;
                    if (turn) {
                        // the ancestors above still add their accept sums
                        for (int al = lev + 1; al < depth; ++al)
                            if ((i_leaf >> al) & 1) T_acc += lsw[al * TS_N + TS_ACC];
                        abort = true;
                        turning = 1;
                        break;
                    }
                    if (!take2) { T_pq = A_pq; T_pE = ls[TS_E]; T_plogp = ls[TS_LOGP]; T_pu = ls[TS_U]; T_pw = ls[TS_W]; }
                    T_ls = ls_new;
                    T_ps = psum;
                    T_lp = A_lp;
                    lev += 1;
                }
                if (abort) { sub_acc = T_acc; done = true; break; }
                if (lev < depth) {
                    // wait for the right sibling
                    if (lev == 0) { L0_lp = T_lp; L0_rp = cur.p; L0_ps = T_ps; L0_pq = T_pq; }
                    else {
                        sb[(size_t)(4 * lev + 0) * 64] = T_lp; sb[(size_t)(4 * lev + 1) * 64] = cur.p;
                        sb[(size_t)(4 * lev + 2) * 64] = T_ps; sb[(size_t)(4 * lev + 3) * 64] = T_pq;
                    }
                    double *ls = lsw + lev * TS_N;
                    if (lane == 0) { ls[TS_LS] = T_ls; ls[TS_ACC] = T_acc; ls[TS_E] = T_pE; ls[TS_LOGP] = T_plogp; ls[TS_U] = T_pu; ls[TS_W] = T_pw; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    sub_acc = T_acc;  // the whole subtree of this doubling is complete
                }
            }
            // (the first state of the completed subtree: the first leaf of the doubling)
            depth += 1;
            accept_sum += sub_acc;
            n_prop += (int)sub_n;
            if (err) break;
            if (diverging || turning) {
                // Tree.extend returns before touching the ends' p_sum (nuts.py:71-73); the new end replaces the old one only for a complete subtree
                break;
            }
            // ---- Tree.extend after a complete subtree: nuts.py:75-103 ----
            // first and last states of the new subtree: the first leaf follows the old edge, the last one is `cur`
            if (dir > 0) right = cur; else left = cur;
            if (logbern(T_ls - log_size)) { prop_q = T_pq; prop_u = T_pu; prop_w = T_pw; prop_E = T_pE; prop_logp = T_plogp; }
            log_size = tn_logaddexp(log_size, T_ls);
            p_sum += T_ps;  // :86 (in place: the aliases below see the new value)
            bool turn = (tn_wsum(p_sum * (var * left.p)) <= 0.) || (tn_wsum(p_sum * (var * right.p)) <= 0.);
            {
                // leftmost / rightmost halves: nuts.py:56-69
                const double sub_first_p = T_lp, sub_last_p = cur.p;
                double lm_begin_p, lm_end_p, rm_begin_p, rm_end_p, lm_ps, rm_ps;
                if (dir > 0) {
                    lm_begin_p = old_left.p; lm_end_p = old_right.p; rm_begin_p = sub_first_p; rm_end_p = sub_last_p;
                    lm_ps = p_sum; rm_ps = T_ps;
                } else {
                    lm_begin_p = sub_last_p; lm_end_p = sub_first_p; rm_begin_p = old_left.p; rm_end_p = old_right.p;
                    lm_ps = T_ps; rm_ps = p_sum;
                }
                const double t1 = lm_ps + rm_begin_p, t2 = lm_end_p + rm_ps;
                turn = turn || (tn_wsum(t1 * (var * lm_begin_p)) <= 0.) || (tn_wsum(t1 * (var * rm_begin_p)) <= 0.);
                turn = turn || (tn_wsum(t2 * (var * lm_end_p)) <= 0.) || (tn_wsum(t2 * (var * rm_end_p)) <= 0.);
            }
            turning = turn ? 1 : 0;
            if (turning) break;
        }
        if (err) break;
        // ---- iteration end: base_hmc.py:252-262 ----
        const double accept_stat = accept_sum / (double)n_prop;
        if (warm && a.cfg.adapt_step_size) {  // step_size.py:31-45
            const double wgt = 1. / (count + a.cfg.t_0);
            hbar = ((1. - wgt) * hbar + wgt * (a.cfg.target_accept - accept_stat));
            log_step = smu - hbar * sqrt(count) / a.cfg.gamma;
            const double mk = exp(-a.cfg.k * log(count));
            log_bar = mk * log_step + (1. - mk) * log_bar;
            count += 1.;
        }
        qc = prop_q;
        u_cur = prop_u;
        const int orow = i_iter - a.iter_out0;
        if (orow >= 0 && orow < a.n_out) {
            if (lane == 0) {
                double *st = a.stats + ((size_t)chain * a.n_out + orow) * BFHIP_STAT_STRIDE;
                st[BFHIP_NS_LOGP] = prop_logp;
                st[BFHIP_NS_ENERGY] = prop_E;
                st[BFHIP_NS_TREE_DEPTH] = (double)depth;
                st[BFHIP_NS_TREE_SIZE] = (double)n_prop;
                st[BFHIP_NS_MEAN_TREE_ACCEPT] = accept_stat;
                st[BFHIP_NS_STEP_SIZE] = exp(log_step);
                st[BFHIP_NS_STEP_SIZE_BAR] = exp(log_bar);
                st[BFHIP_NS_WARMUP] = warm ? 1. : 0.;
                st[BFHIP_NS_ENERGY_CHANGE] = prop_E - start.energy;
                st[BFHIP_NS_MAX_ENERGY_CHANGE] = max_de;
                st[BFHIP_NS_DIVERGING] = (double)diverging;
                double *tt = a.stats_t + ((size_t)chain * a.n_out + orow) * 2;
                tt[0] = prop_u;
                tt[1] = prop_w;
            }
            if (in) a.samples[((size_t)chain * a.n_out + orow) * d + lane] = qc;
        }
        if (warm && a.cfg.adapt_metric) {  // QuadMetricDiagAdapt.update: metrics.py:186-211
            const long delta = (long)(n_samples - prev_upd);
            double fm = in ? vecp[BFHIP_VEC_FG_MEAN * d + lane] : 0., fr = in ? vecp[BFHIP_VEC_FG_RAW * d + lane] : 0.;
            double bm = in ? vecp[BFHIP_VEC_BG_MEAN * d + lane] : 0., br = in ? vecp[BFHIP_VEC_BG_RAW * d + lane] : 0.;
            fg_n += 1.; bg_n += 1.;
            double od = qc - fm; fm += od / fg_n; fr += 1. * od * (qc - fm);
            od = qc - bm; bm += od / bg_n; br += 1. * od * (qc - bm);
            if ((delta + 1) % (long)a.cfg.update_window == 0) {
                if (in) { var = fr / fg_n; vecp[BFHIP_VEC_VAR * d + lane] = var; }
            }
            if ((double)delta >= adapt_window) {
                fm = bm; fr = br; bm = 0.; br = 0.;
                fg_n = bg_n; bg_n = 10.; prev_upd = n_samples;
                if (a.cfg.doubling) adapt_window *= 2.;
            }
            n_samples += 1.;
            if (in) {
                vecp[BFHIP_VEC_FG_MEAN * d + lane] = fm; vecp[BFHIP_VEC_FG_RAW * d + lane] = fr;
                vecp[BFHIP_VEC_BG_MEAN * d + lane] = bm; vecp[BFHIP_VEC_BG_RAW * d + lane] = br;
            }
        }
        i_iter += 1;
    }
    // ---- write the chain state back ----
    if (in) vecp[BFHIP_VEC_Q * d + lane] = qc;
    if (lane == 0) {
        for (int k = 0; k < 4; ++k) a.rng[(size_t)chain * 4 + k] = rs[k];
        scp[BFHIP_SC_LOG_STEP] = log_step; scp[BFHIP_SC_LOG_BAR] = log_bar; scp[BFHIP_SC_HBAR] = hbar; scp[BFHIP_SC_COUNT] = count;
        scp[BFHIP_SC_FG_N] = fg_n; scp[BFHIP_SC_BG_N] = bg_n; scp[BFHIP_SC_N_SAMPLES] = n_samples;
        scp[BFHIP_SC_PREV_UPDATE] = prev_upd; scp[BFHIP_SC_ADAPT_WINDOW] = adapt_window;
        scp[BFHIP_SC_I_ITER] = (double)i_iter; scp[BFHIP_SC_ERROR] = (double)err;
        a.tu[chain] = u_cur;
        if (a.n_leapfrog && nlf) atomicAdd(a.n_leapfrog, nlf);
    }
}

static int g_tnuts_wpb = [] { const char *e = getenv("BFHIP_TNUTS_WPB"); return e ? atoi(e) : 0; }();
extern "C" void bfhip_debug_tnuts_wpb(int v) { g_tnuts_wpb = v; }  // test / tuning hook: waves per workgroup (8, 16; 0: automatic)

extern "C" int bfhip_tnuts_run(bfhip_ctx *ctx, const bfhip_sampler_config *cfg, const bfhip_tempering *tp, int n_chain, int iter_end,
                               uint64_t *rng, double *sc, double *vec, double *u, int iter_out0, int n_out, double *samples,
                               double *stats, double *stats_t, unsigned long long *n_leapfrog) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || !cfg || !tp || n_chain < 0) return bf_set_error(BFHIP_ERR_ARG, "bfhip_tnuts_run: invalid argument");
    if (!ctx->has_model) return bf_set_error(BFHIP_ERR_STATE, "bfhip_tnuts_run: no density uploaded");
    if (ctx->model.pld.on) return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_tnuts_run: not implemented for the pipeline density");
    if (n_chain == 0) return 0;
    if (!rng || !sc || !vec || !u || !tp->base_S || !tp->base_lin || n_out < 0 || (n_out > 0 && (!samples || !stats || !stats_t)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_tnuts_run: NULL array");
    if (cfg->max_treedepth < 1 || cfg->max_treedepth > BFHIP_MAX_TREEDEPTH || !(cfg->max_change > 0.) || cfg->update_window < 1)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_tnuts_run: invalid sampler configuration");
    const DevModel &m = ctx->model;
    if (!bf_model_plain(m) || m.DP > 64 || cfg->full_metric)
        return bf_set_error(BFHIP_ERR_UNSUPPORTED, "bfhip_tnuts_run: the tempered sampler covers the common surrogate (linear + quadratic "
                                                   "configs with the bound, no transform / scaling / decay / cubic) at d <= 64 with the diagonal metric");
    // Waves (= chains) per workgroup.  The three staged matrices take 96 KB of LDS, so a CU holds ONE workgroup: with 8 waves
    // of 256 registers 2048 chains run at a time and 4096 chains take two rounds; 16 waves of 128 registers (spilling ~50)
    // keep all of them resident.  BFHIP_TNUTS_WPB overrides (tuning).
    const int forced = g_tnuts_wpb;
    const int wpb = forced == 8 || forced == 16 ? forced : (n_chain > 8 * ctx->n_cu ? 16 : 8);
    const size_t need = (size_t)((n_chain + 15) / 16 * 16) * (4 * TN_MAXL) * 64 * sizeof(double);
    if (ctx->scratch_bytes < need) {
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) BF_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = NULL;
        ctx->scratch_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    TnutsArgs a;
    a.cfg = *cfg;
    a.n_chain = n_chain; a.iter_end = iter_end; a.iter_out0 = iter_out0; a.n_out = n_out; a.d = m.d;
    a.rng = rng; a.sc = sc; a.vec = vec; a.tu = u; a.samples = samples; a.stats = stats; a.stats_t = stats_t;
    a.n_leapfrog = n_leapfrog;
    a.scratch = (double *)ctx->scratch;
    a.base_S = tp->base_S; a.base_lin = tp->base_lin; a.base_c0 = tp->base_c0; a.logxi = tp->logxi;
    const size_t lds = ((size_t)3 * 4096 + wpb * 64 + wpb * TN_MAXL * TS_N) * sizeof(double);
    if (wpb == 16) {
        auto k = bf_tnuts_kernel<16>;
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((n_chain + 15) / 16), dim3(64 * 16), lds, ctx->stream, m, a);
    } else {
        auto k = bf_tnuts_kernel<8>;
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((n_chain + 7) / 8), dim3(64 * 8), lds, ctx->stream, m, a);
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
