// bfhip_tnuts.hip -- tempered NUTS: BaseTHMC.astep (samplers/hmc_utils/base_hmc.py:220-262) around
// the NUTS tree (samplers/nuts.py:21-217, TTree: samplers/tnuts.py:15-41) with TCpuLeapfrogIntegrator
// (samplers/hmc_utils/integration.py:98-222): the state carries a tempering coordinate u with momentum v, the potential
// is beta(u) phi + (1 - beta(u)) psi + U(u) with phi = -logp of the surrogate target and psi = -(logp of a base density
// + log xi), and every leapfrog step evaluates both densities twice (mid-point gradients, end-point values).
//
// Layout: ONE WAVE PER CHAIN, lane = dimension (d <= 64), eight waves of 256 registers per workgroup (two workgroups per CU;
// sixteen waves of 128 registers spilled 275 of them), and the three matrix-vector products of an evaluation -- S q,
// H (q - mu), S_b q -- SHARED by the workgroup's chains on FP64 MFMA tiles, as in the sliced sampler kernel: every evaluation is
// a RENDEZVOUS of the workgroup (the chains put their point into the B operand, column = wave; barrier; wave t < W runs row tile
// t of S and of S_b -- the same operand, two independent chains of d / 4 v_mfma_f64_16x16x4_f64 -- and wave 4 + t row tile t of
// H, with the A fragments they keep in registers for the whole launch; barrier; every chain reads its column).  The tree logic between two evaluations is
// each wave's own (the reference's recursion, flattened); a chain that has finished its iterations keeps answering the
// rendezvous until no chain of the workgroup is active.  (Until round 4 every wave ran its own 3 x 64 broadcast-FMA steps per
// evaluation from matrices staged in LDS: 1.05 x 10^8 tempered steps/s at 4096 chains x 64-d.)  Outside the bound the target
// follows by linearity (bfhip_oob.h), as in the other sampler kernels.  Target: the common surrogate (linear + quadratic
// configs with the extrapolation bound, no transform / scaling / decay / cubic).  Base: a quadratic log-density without
// bound (e.g. the Gaussian approximation of the posterior).  Draws are consumed in the recursion's post-order, as in the
// other sampler kernels, so a chain reproduces the CPU oracle for the same xoshiro stream.
#include <cmath>
#include "bfhip_common.h"
#include "bfhip_eval.h"
#include "bfhip_sampler_defs.h"
#include "bfhip_wave.h"
#include "bfhip_oob.h"

#include "bfhip_tnuts.h"

#define TN_XS 65   // row stride of the B operands and of the results (doubles)

#define TN_WAVES 8
// TR: the target lives behind the constraint transform (Density.input_scales / hard_bounds: density.py:92-140, 747-750) -- the
// surrogate is evaluated at x(q), its gradient gets the chain-rule factor and the log-Jacobian term; the base density stays in
// the sampler's space (base_hmc.py:227-231 evaluates both at q).  DEC: the decay penalty of the target (density.py:740-746), a
// fourth product H_d (x - mu_d) on the waves that run H.  (Round 5: every GBS example of the reference has bounds or decay.)
template <int W, bool TR = false, bool DEC = false>   // row tiles of the matrices: the padded dimension is 16 W (W = 1, 2, 4), known at compile time
__global__ __launch_bounds__(64 * TN_WAVES) void bf_tnuts_kernel(DevModel m, TnutsArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int d = a.d;
    constexpr int NXB = 4, NGB = 4;          // (regions 2 and 3 are used with TR / DEC only)
    double *XB = lds;                        // [4][16][TN_XS] B operands: x | x - mu | q (base; = x without TR) | x - mu_decay  (k-step s, lane l: dimension 4 s + (l >> 4) of chain l & 15)
    double *GB = XB + NXB * 16 * TN_XS;      // [4][16][TN_XS] results: S x | H (x - mu) | S_b q | H_d (x - mu_d), [matrix][chain][dimension]
    double *LSC = GB + NGB * 16 * TN_XS;     // [TN_WAVES][TN_MAXL][TS_N]
    int *flags = (int *)(LSC + TN_WAVES * TN_MAXL * TS_N);   // [2] some chain of the workgroup is active (by rendezvous parity)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chain = blockIdx.x * a.cpg + w;
    const bool real = w < a.cpg && chain < a.n_chain;
    constexpr int NS = 4 * W, DPW = 16 * W;
    // A operands of this wave's jobs, in registers for the whole launch: waves 0..3 row tile w of S (afr) and of S_b (afb), waves
    // 4..7 row tile w - 4 of H (afr)
    const int jt = w & 3;                 // row tile
    const bool has_job = jt < W, job_h = w >= 4;
    double afr[16], afb[16];   // (afb: S_b on waves 0-3, H_decay on waves 4-7)
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        double v = 0., vb = 0.;
        if (has_job && s2 < NS) {
            v = (job_h ? m.Hf : m.Sf)[((size_t)jt * NS + s2) * 64 + lane];
            const int row = 16 * jt + (lane & 15), col = 4 * s2 + (lane >> 4);
            if (!job_h && row < d && col < d) vb = a.base_S[(size_t)row * d + col];
            if (DEC && job_h) vb = m.Hdf[((size_t)jt * NS + s2) * 64 + lane];
        }
        afr[s2] = v;
        afb[s2] = vb;
    }
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < (NXB + NGB) * 16 * TN_XS; i += 64 * TN_WAVES) XB[i] = 0.;   // (XB and GB: the columns without a chain stay zero)
    __syncthreads();
    double *lsw = LSC + w * (TN_MAXL * TS_N);
    const bool in = lane < d;
    const double c_lin = in ? m.pd[PD_LIN * DPW + lane] : 0., c_mu = in ? m.pd[PD_MU * DPW + lane] : 0.;
    const double c_smu = in ? m.pd[PD_SMU * DPW + lane] : 0.;
    const double b_lin = in ? a.base_lin[lane] : 0.;
    const double c_dmu = (DEC && in) ? m.pd[PD_DMU * DPW + lane] : 0.;
    const int c_kind = (TR && in) ? (int)m.pd[PD_KIND * DPW + lane] : 0;
    const double c_lo = (TR && in) ? m.pd[PD_LO * DPW + lane] : 0., c_rg = (TR && in) ? m.pd[PD_RG * DPW + lane] : 1.;
    // One rendezvous of the workgroup: this wave's point (active: it has one) -> S q, H (q - mu), S_b q of its chain.  Returns
    // false when no chain of the workgroup is active any more (the same answer in every wave).
    int n_x = 0;
    auto exchange = [&](bool active, double x, double xm, double qb, double xd, double &sx, double &hv, double &bx, double &dgr) -> bool {
        const int par = n_x & 1;
        n_x += 1;
        if (lane < DPW) {
            const int xi = (lane >> 2) * TN_XS + w + 16 * (lane & 3);
            XB[xi] = x;
            XB[16 * TN_XS + xi] = xm;
            if constexpr (TR) XB[2 * 16 * TN_XS + xi] = qb;
            if constexpr (DEC) XB[3 * 16 * TN_XS + xi] = xd;
        }
        if (active && lane == 0) flags[par] = 1;
        __syncthreads();  // R1
        const bool any = rfl(flags[par]) != 0;
        if (threadIdx.x == 0) flags[par ^ 1] = 0;
        if (any && has_job) {
            // eight columns: two v_mfma_f64_4x4x4_4b per k-step (columns 0-3 and 4-7; four 4-row blocks of the tile against the
            // same four columns: A lane 16 k + m as for the 16 x 16 x 4 tile, B lane 16 k + 4 b + n reads column n, D lane
            // 16 i + 4 b + n is row 4 b + i of column n) -- 36 against 64 cycles of the FP64 pipe, the same sequential sum per entry
            const double *Xq = XB + (job_h ? 16 * TN_XS : 0) + (lane & ~15) + (lane & 3);
            // the second chain of the wave: S_b against q (region 2 with the transform, region 0 = x = q without) on waves 0-3,
            // H_decay against x - mu_decay (region 3) on waves 4-7
            const bool second = !job_h || DEC;
            const double *Xs = XB + (job_h ? 3 : (TR ? 2 : 0)) * 16 * TN_XS + (lane & ~15) + (lane & 3);
            double a0 = 0., a1 = 0., b0 = 0., b1 = 0.;
#pragma unroll
            for (int c0 = 0; c0 < 16; c0 += 4) {
                if (c0 < NS) {
                    double x0[4], x1[4], y0[4], y1[4];
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) { x0[s2] = Xq[(c0 + s2) * TN_XS]; x1[s2] = Xq[(c0 + s2) * TN_XS + 4]; }
                    if constexpr (TR || DEC) {
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) { y0[s2] = Xs[(c0 + s2) * TN_XS]; y1[s2] = Xs[(c0 + s2) * TN_XS + 4]; }
                    } else {
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) { y0[s2] = x0[s2]; y1[s2] = x1[s2]; }
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(afr[c0 + s2], x0[s2], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(afr[c0 + s2], x1[s2], a1, 0, 0, 0);
                        if (second) {
                            b0 = __builtin_amdgcn_mfma_f64_4x4x4f64(afb[c0 + s2], y0[s2], b0, 0, 0, 0);
                            b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(afb[c0 + s2], y1[s2], b1, 0, 0, 0);
                        }
                    }
                }
            }
            const int col = lane & 3, row = 16 * jt + 4 * ((lane >> 2) & 3) + (lane >> 4);
            GB[((job_h ? 1 : 0) * 16 + col) * TN_XS + row] = a0;
            GB[((job_h ? 1 : 0) * 16 + 4 + col) * TN_XS + row] = a1;
            if (second) {
                GB[((job_h ? 3 : 2) * 16 + col) * TN_XS + row] = b0;
                GB[((job_h ? 3 : 2) * 16 + 4 + col) * TN_XS + row] = b1;
            }
        }
        __syncthreads();  // R2
        const bool rd = lane < DPW;
        sx = rd ? GB[(0 * 16 + w) * TN_XS + lane] : 0.;
        hv = rd ? GB[(1 * 16 + w) * TN_XS + lane] : 0.;
        bx = rd ? GB[(2 * 16 + w) * TN_XS + lane] : 0.;
        dgr = (DEC && rd) ? GB[(3 * 16 + w) * TN_XS + lane] : 0.;
        return any;
    };
    // phi, dphi, psi, dpsi at q (this lane's coordinate): integration.py:180-181 / base_hmc.py:227-231
    auto potentials = [&](double q, double &phi, double &dphi, double &psi, double &dpsi) {
        // target surrogate with its bound (modules/poly.py:466-503), behind the constraint transform when there is one
        double x = q, jac = 1., gj = 0., logdet_l = 0.;
        if constexpr (TR) {
            double J, J2;
            bf_to_original(q, c_kind, c_lo, c_rg, x, J, J2);
            if (in) {
                logdet_l = log(fabs(J));
                jac = J;
                gj = J2 / J;
            } else {
                x = 0.;
            }
        }
        const double xm = in ? x - c_mu : 0.;
        const double xd = (DEC && in) ? x - c_dmu : 0.;
        double sx, hv, bx, dgr;
        (void)exchange(true, in ? x : 0., xm, in ? q : 0., xd, sx, hv, bx, dgr);
        double gn = sx + c_lin;
        // the evaluation's sums in one reduction: target value, bound, base value, the two sums of the extrapolation outside the
        // bound (bfhip_oob.h), the log-Jacobian and the decay term's radius
        const double sv = sx - c_smu, gmu = c_smu + c_lin;
        double r7[7] = {in ? __builtin_fma(0.5 * x, sx, c_lin * x) : 0., xm * hv, in ? __builtin_fma(0.5 * q, bx, b_lin * q) : 0.,
                        xm * gmu, xm * sv, logdet_l, xd * dgr};
        wave_sum_n<(TR || DEC) ? 7 : 5>(reinterpret_cast<double (&)[(TR || DEC) ? 7 : 5]>(r7));
        double f = m.c0 + r7[0];
        const double beta = sqrt(r7[1]);
        if (beta > m.alpha) {
            const BfOob o = bf_oob_scalars(m.alpha, m.inv_alpha, m.f_mu, m.f_poly_mu, beta, r7[3], r7[4]);
            f = o.f;
            gn = bf_oob_grad(o, gmu, sv, hv);
        }
        gn = gn * jac;   // chain rule (module.py:226, density.py:558); 1 without the transform
        if constexpr (DEC) {  // density.py:740-746 (the decay gradient is added in the original space, as in the reference)
            f -= m.decay_gamma * bf_clip0(r7[6] - m.decay_alpha2);
            if (r7[6] > m.decay_alpha2) gn -= 2. * m.decay_gamma * dgr;
        }
        if constexpr (TR) {   // density.py:747-750
            f += r7[5];
            gn += gj;
        }
        phi = rfl(-f);   // (wave-uniform values go back to scalar registers: FP64 arithmetic leaves them in vector ones)
        dphi = in ? -gn : 0.;
        // base: c0 + lin.x + x.S_b x / 2, plus log xi
        const double fb = a.base_c0 + r7[2];
        psi = rfl(-(fb + a.logxi));
        dpsi = in ? -(bx + b_lin) : 0.;
    };

    if (real) {
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
        double u_cur = rfl(a.tu[chain]);
        unsigned long long nlf = 0;
        auto uni = [&]() { return bf_u01(bf_xoshiro_next(rs)); };
        auto logbern = [&](double l) -> bool {  // nuts.py:200-203
            if (l != l) err = 2;
            return log(uni()) < l;
        };

        // one tempered leapfrog step from (q, p, u, vt): integration.py:153-222
        // (weight: phi - psi of the state; the importance weight delta / expm1(delta), base_hmc.py:227-231, is taken once per
        // iteration, for the proposal that is kept)
        struct TS { double q, p, u, vt, weight, energy, logp; };
        auto finish_state = [&](TS &s, double phi, double psi) {
            const double kin = tn_wsum(s.p * (var * s.p));
            const double ope = 1 + exp(-s.u), beta = 1 / ope, pot = s.u + 2 * log(ope);   // t_beta, t_pot: one exponential
            s.energy = rfl((beta * phi + (1 - beta) * psi + pot) + (0.5 * kin + s.vt * s.vt / 2));
            s.logp = rfl(-phi);
            s.weight = rfl(phi - psi);
        };
        auto t_step = [&](const TS &s0, double eps) -> TS {
            TS s = s0;
            const double dt = 0.5 * eps;
            double phi, dphi, psi, dpsi;
            s.u = rfl(s.u + s.vt * dt);
            s.q += dt * (var * s.p);
            potentials(s.q, phi, dphi, psi, dpsi);
            // beta(u) = 1 / (1 + e), beta'(u) = e / (1 + e)^2, U'(u) = (e^u - 1) / (e^u + 1) = (1 - e) / (1 + e) with e = exp(-u): one
            // exponential and one division (the reference's forms to rounding, integration.py:186-200)
            const double e = exp(-s.u), beta = 1 / (1 + e), dbeta = e * beta * beta, dU = (1 - e) * beta;
            s.vt = rfl(s.vt + -(dbeta * (phi - psi) + dU) * eps);
            s.p += eps * -(beta * dphi + (1 - beta) * dpsi);
            s.u = rfl(s.u + s.vt * dt);
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
                start.vt = rfl(sqrt(-2. * log(u1)) * cs);
            }
            start.u = u_cur;
            {
                double phi, dphi, psi, dpsi;
                potentials(start.q, phi, dphi, psi, dpsi);
                finish_state(start, phi, psi);
            }
            if (!(fabs(start.energy) <= 1.7976931348623157e308)) { err = 1; break; }
            const double eps0 = rfl(exp(warm ? log_step : log_bar));
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
                    double dE = rfl(nxt.energy - start.energy);
                    if (dE != dE) dE = INFINITY;
                    if (fabs(dE) > fabs(max_de)) max_de = dE;
                    if (!(fabs(dE) < a.cfg.max_change)) {
                        diverging = 1;
                        // the stub subtree: ancestors still add their left halves' accept sums (nuts.py:173)
                        for (int al = 0; al < depth; ++al)
                            if ((i_leaf >> al) & 1) sub_acc = rfl(sub_acc + (al == 0 ? lsw[TS_ACC] : lsw[al * TS_N + TS_ACC]));
                        done = true;
                        break;
                    }
                    cur = nxt;
                    T_lp = nxt.p; T_ps = nxt.p;
                    T_pq = nxt.q; T_pu = nxt.u; T_pw = nxt.weight; T_pE = nxt.energy; T_plogp = nxt.logp;
                    T_ls = -dE;
                    { const double pa = rfl(exp(-dE)); T_acc = pa > 1. ? 1. : pa; }
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
                        bool turn;
                        if (lev >= 1) {  // with the sub-span checks for depth > 1 (nuts.py:154-161): six sums, one reduction
                            const double ps1 = A_ps + T_lp;
                            const double ps2 = A_rp + T_ps;
                            double r6[6] = {psum * A_lv, psum * (var * cur.p), ps1 * A_lv, ps1 * (var * T_lp), ps2 * A_rv, ps2 * (var * cur.p)};
                            wave_sum_n<6>(r6);
                            turn = (r6[0] <= 0.) || (r6[1] <= 0.) || (r6[2] <= 0.) || (r6[3] <= 0.) || (r6[4] <= 0.) || (r6[5] <= 0.);
                        } else {
                            double r2[2] = {psum * A_lv, psum * (var * cur.p)};
                            wave_sum_n<2>(r2);
                            turn = (r2[0] <= 0.) || (r2[1] <= 0.);
                        }
                        const double acc_l = rfl(ls[TS_ACC]), ls_l = rfl(ls[TS_LS]);
                        const double ls_new = rfl(tn_logaddexp(ls_l, T_ls));
                        const bool take2 = logbern(T_ls - ls_new);  // :164 (drawn even when this merge turns)
                        T_acc = rfl(acc_l + T_acc);
                        if (turn) {
                            // the ancestors above still add their accept sums
                            for (int al = lev + 1; al < depth; ++al)
                                if ((i_leaf >> al) & 1) T_acc = rfl(T_acc + lsw[al * TS_N + TS_ACC]);
                            abort = true;
                            turning = 1;
                            break;
                        }
                        if (!take2) { T_pq = A_pq; T_pE = rfl(ls[TS_E]); T_plogp = rfl(ls[TS_LOGP]); T_pu = rfl(ls[TS_U]); T_pw = rfl(ls[TS_W]); }
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
                accept_sum = rfl(accept_sum + sub_acc);
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
                log_size = rfl(tn_logaddexp(log_size, T_ls));
                p_sum += T_ps;  // :86 (in place: the aliases below see the new value)
                bool turn;
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
                    double r6[6] = {p_sum * (var * left.p), p_sum * (var * right.p), t1 * (var * lm_begin_p), t1 * (var * rm_begin_p),
                                    t2 * (var * lm_end_p), t2 * (var * rm_end_p)};
                    wave_sum_n<6>(r6);
                    turn = (r6[0] <= 0.) || (r6[1] <= 0.) || (r6[2] <= 0.) || (r6[3] <= 0.) || (r6[4] <= 0.) || (r6[5] <= 0.);
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
                    tt[1] = (prop_w == 0) ? 1. : prop_w / expm1(prop_w);
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
    // the chains of this wave's workgroup that are still running need its matvec job (and the barriers)
    {
        double t0, t1, t2, t3;
        while (exchange(false, 0., 0., 0., 0., t0, t1, t2, t3)) { }
    }
}


extern "C" int bfhip_tnuts_run(bfhip_ctx *ctx, const bfhip_sampler_config *cfg, const bfhip_tempering *tp, int n_chain, int iter_end,
                               uint64_t *rng, double *sc, double *vec, double *u, int iter_out0, int n_out, double *samples,
                               double *stats, double *stats_t, unsigned long long *n_leapfrog) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || !cfg || !tp || n_chain < 0) return bf_set_error(BFHIP_ERR_ARG, "bfhip_tnuts_run: invalid argument");
    if (!ctx->has_model) return bf_set_error(BFHIP_ERR_STATE, "bfhip_tnuts_run: no density uploaded");
    if (n_chain == 0) return 0;
    if (!rng || !sc || !vec || !u || !tp->base_S || !tp->base_lin || n_out < 0 || (n_out > 0 && (!samples || !stats || !stats_t)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_tnuts_run: NULL array");
    if (cfg->max_treedepth < 1 || cfg->max_treedepth > BFHIP_MAX_TREEDEPTH || !(cfg->max_change > 0.) || cfg->update_window < 1)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_tnuts_run: invalid sampler configuration");
    if (cfg->full_metric && !cfg->metric_mat) return bf_set_error(BFHIP_ERR_ARG, "bfhip_tnuts_run: full_metric needs metric_mat");
    const DevModel &m = ctx->model;
    // the tuned instantiations: the common surrogate (linear + quadratic configs with the bound; constraint transform and decay
    // optional) at d <= 64 with the diagonal metric.  Everything else -- cubic configs, d = 128, device-side input scaling, the
    // Gaussian link, the pipeline density, the full-rank metric -- runs on the generic kernel (bfhip_tnuts_gen.hip)
    const bool common = m.has_quad && m.use_bound && !m.has_su && !m.has_cubic && !m.has_link && !m.pld.on;
    const bool generic = !common || m.DP > 64 || cfg->full_metric || bf_tune().tnuts_generic;
    // Chains per workgroup (a workgroup is always eight waves, two workgroups per CU: the waves without a chain run matvec jobs
    // only): eight, or four when that spreads few chains over more CUs.  BFHIP_TNUTS_WPB / bfhip_debug_set("tnuts_wpb") override.
    const int forced = bf_tune().tnuts_wpb;
    const int cpg = (forced == 4 || forced == 8) ? forced : (n_chain > 8 * ctx->n_cu ? 8 : 4);
    const size_t need = (size_t)((n_chain + 15) / 16 * 16) * (4 * TN_MAXL) * 64 * sizeof(double);
    if (ctx->scratch_bytes < need) {
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) BF_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = NULL;
        ctx->scratch_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    if (generic) {
        TnutsArgs g;
        g.cfg = *cfg;
        g.n_chain = n_chain; g.iter_end = iter_end; g.iter_out0 = iter_out0; g.n_out = n_out; g.d = m.d;
        g.rng = rng; g.sc = sc; g.vec = vec; g.tu = u; g.samples = samples; g.stats = stats; g.stats_t = stats_t;
        g.n_leapfrog = n_leapfrog;
        g.scratch = NULL;
        g.base_S = tp->base_S; g.base_lin = tp->base_lin; g.base_c0 = tp->base_c0; g.logxi = tp->logxi;
        g.cpg = cpg;
        return bf_tnuts_gen_launch(ctx, g, cfg->full_metric ? (const double *)cfg->metric_mat : NULL);
    }
    TnutsArgs a;
    a.cfg = *cfg;
    a.n_chain = n_chain; a.iter_end = iter_end; a.iter_out0 = iter_out0; a.n_out = n_out; a.d = m.d;
    a.rng = rng; a.sc = sc; a.vec = vec; a.tu = u; a.samples = samples; a.stats = stats; a.stats_t = stats_t;
    a.n_leapfrog = n_leapfrog;
    a.scratch = (double *)ctx->scratch;
    a.base_S = tp->base_S; a.base_lin = tp->base_lin; a.base_c0 = tp->base_c0; a.logxi = tp->logxi;
    a.cpg = cpg;
    const size_t lds = ((size_t)8 * 16 * TN_XS + TN_WAVES * TN_MAXL * TS_N + 2) * sizeof(double);
    const bool tr = m.has_transform != 0, dec = m.use_decay != 0;
    void (*k)(DevModel, TnutsArgs) = NULL;
#define TN_PICK(Wv) (tr ? (dec ? bf_tnuts_kernel<Wv, true, true> : bf_tnuts_kernel<Wv, true, false>) : (dec ? bf_tnuts_kernel<Wv, false, true> : bf_tnuts_kernel<Wv, false, false>))
    k = m.DP == 64 ? TN_PICK(4) : (m.DP == 32 ? TN_PICK(2) : TN_PICK(1));
#undef TN_PICK
    if (lds > 64 * 1024) BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((n_chain + cpg - 1) / cpg), dim3(64 * TN_WAVES), lds, ctx->stream, m, a);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
