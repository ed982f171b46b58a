// bfhip_nuts_pipe.h -- software-pipelined NUTS transitions for the common surrogate at d <= 64 (gfx950).
// Included by bfhip_sampler.hip (uses its helpers: rfl, wave_sum_n, the out-of-line libm wrappers, SamplerArgs, enums).
//
// Same decomposition as bf_sampler_kernel (one wave per chain, 16 chains per workgroup, the batched matvecs S X^T and
// H (X - mu)^T of the group on v_mfma_f64_16x16x4_f64, two workgroup barriers per trip) and the SAME arithmetic per
// chain, in the same order: a chain's samples and statistics are bit-identical to bf_sampler_kernel's.  What changes is
// WHEN the tree bookkeeping runs:
//
//   * bf_sampler_kernel gives every unit of bookkeeping (leaf logic, a merge level, the end of a doubling, the pieces
//     of the iteration end) a trip of its own or the tail of one, so a 7-leaf iteration takes 15 trips, each with its
//     two workgroup barriers, flag exchange and loop overhead, and seven of them without any gradient work.
//   * Here the bookkeeping of leaf n is DEFERRED into trip n+1, between the two barriers, next to the MFMAs of that
//     trip's gradient tiles.  The leapfrog step of trip n+1 therefore starts BEFORE leaf n has been accounted for: it
//     is speculative.  The speculation is exact whenever the tree goes on: inside a doubling the next leaf continues
//     from the current one, and at the end of a doubling the direction of the next one is read ahead from the chain's
//     random stream (the number of draws the pending bookkeeping will consume is known: one per merge level and one
//     for the swap, samplers/nuts.py:163-167,81-83, then the direction, :210).  When the pending bookkeeping ends the
//     tree (U-turn, divergence, depth limit) the evaluation in flight is dropped; that is one wasted evaluation per
//     iteration, in a trip the group runs anyway.  The same bookkeeping pass then does the iteration-end work (step
//     size, statistics, sample, metric window, next momentum) and starts the next iteration WITHOUT the evaluation
//     that opens it in the reference (compute_state, base_hmc.py:70): its point is the proposal, a leaf whose value
//     and gradient the tree has computed; the gradient travels with the proposal through the merges (TPg, L0g, one
//     scratch vector per stack level).  A 7-leaf iteration takes 8 trips.
//   * The subtree ends, the proposal, p_sum and stack level 1 live in LDS (12 vectors per chain); only the deeper
//     stack levels go to global scratch.  There is no prefetch buffer and no tail path (the row-major matrices of the
//     tail path do not fit next to the tree vectors).
//   * What does NOT happen: the bookkeeping does not hide behind the MFMAs.  On gfx950 v_mfma_f64_16x16x4_f64 and
//     the FP64 VALU instructions share one pipe (tools/probe/mfma_overlap_probe.hip: a SIMD's time is the SUM of its
//     MFMA cycles and its FP64 VALU cycles, from one wave or from four), so a trip costs its 32 MFMAs per SIMD plus
//     the VALU instructions of its four waves.  The gain over bf_sampler_kernel is the 7 of 15 trips per iteration that
//     no longer exist: +29 % on the headline workload (8.0e8 against 6.2e8 leapfrog steps/s).
//
// E = 1 throughout (lane = dimension, d <= 64).


// scratch slots behind the subtree stack: the gradient at the tree's proposal and at the proposals of the stacked subtrees
// (levels >= 1); they let an iteration start from its predecessor's proposal without evaluating it again
enum { SL_PROPG = SL_STACK + 4 * BFHIP_MAX_TREEDEPTH, SL_PG = SL_PROPG + 1, SL_PIPE_N = SL_PG + BFHIP_MAX_TREEDEPTH };

// DEC = 1: a third matrix, the decay term's (density.py:740-746).  DEC = 2 (round 6): the decay term's matrix and centre ARE the
// bound's -- SurrogateDensity.fit takes both from the same points by the same statements (modules/poly.py:262-276 and
// core/density.py:796-811: mean and inv(cov)), the upload compares the arrays bit for bit -- so H_d (x - mu_d) is the product the
// bound already needs and the radius of the decay term is the bound's: two matrices, with DEC = 1's K-split, i.e. the same job
// per (matrix, row tile): the numbers are DEC = 1's to the last bit.  The K-split is the sliced kernel's for the same model
// (sampler_ksplit: at most 16 jobs), so that the sums associate the same way: at W = 4 twelve jobs of 16 k-steps.
template <int W, int DEC = 0>
struct PipeGeo {
    static constexpr int NMAT = DEC == 1 ? 3 : 2;
    static constexpr int KS = DEC == 1 ? (W == 2 ? 2 : 1) : ((W == 2 || W == 4) ? 2 : 1);  // K-split of the matvec jobs: every wave owns at most one job
    static constexpr int KPJ = (4 * W) / KS, NJOB = NMAT * W * KS, NTL = 12;
    static constexpr int MPS = KPJ > 8 ? 2 : 1;  // MFMAs at each of the eight points of phase B the chain is spread over
    static_assert(NJOB <= 16 && KPJ <= 8 * MPS, "one job per wave, eight MFMA sites");
    static constexpr size_t lds_doubles() {
        using G = SamplerGeo<W>;
        return (size_t)NMAT * G::NS * G::XS + (size_t)16 * BFHIP_MAX_TREEDEPTH * LS_N + 4 + (size_t)16 * CS_N +
               (size_t)16 * NTL * G::DP + (size_t)NMAT * KS * 16 * G::GS;
    }
};

// TR: the density lives behind the constraint transform (Density.input_scales / hard_bounds: density.py:92-140,
// 747-750): the surrogate is evaluated at x(q), its gradient gets the chain-rule factor dx/dq and the log-Jacobian
// term; same arithmetic as the FS = 5 instantiation of bf_sampler_kernel.
// QUAD: at most four chains in the workgroup (wave_layout_cpg) -- the jobs run on v_mfma_f64_4x4x4_4b, four 4-row blocks of
// the tile against the same four columns (lane maps: bf_sampler_kernel's run_jobs), a quarter of the 16-column tile's time
// in the FP64 pipe that the bookkeeping's own FP64 instructions share; the same sequential sum per entry.
// (QUAD = 2: at most eight chains -- two such instructions per k-step, columns 0-3 and 4-7: 35 against 64 cycles of the pipe)
template <int QUAD> struct PipeAcc { typedef d4_t type; };
template <> struct PipeAcc<1> { typedef double type; };
template <> struct PipeAcc<2> { typedef d2_t type; };
__device__ inline d4_t bf_pipe_mfma(double a_, double b_, d4_t c_) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, c_, 0, 0, 0); }
__device__ inline double bf_pipe_mfma(double a_, double b_, double c_) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a_, b_, c_, 0, 0, 0); }

template <int W, bool TR, int DEC = 0, int QUAD = 0>
__global__ __launch_bounds__(1024) void bf_nuts_pipe_kernel(DevModel m, SamplerArgs a) {
    using G = SamplerGeo<W>;
    using PG = PipeGeo<W, DEC>;
    constexpr int DP = G::DP, NS = G::NS, XS = G::XS, GS = G::GS;
    constexpr int KS_P = PG::KS, KPJ_P = PG::KPJ, NJOB_P = PG::NJOB, NTL = PG::NTL, NMAT = PG::NMAT, MPS = PG::MPS;
    constexpr int MAXL = BFHIP_MAX_TREEDEPTH;
    static_assert(DP <= 64, "one dimension per lane");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *XB = lds;                             // [NMAT][NS][XS] B operands: x | x - mu | x - mu_decay
    double *LS = XB + NMAT * NS * XS;             // [16][MAXL][LS_N] per-chain stack scalars
    int *alive = (int *)(LS + 16 * MAXL * LS_N);  // [2] any chain not done | [2] mask of the chains evaluating (by trip parity) | [6] number of
                                                  // waves whose chain has left (tail_stop)
    double *CS = LS + 16 * MAXL * LS_N + 4;       // [16][CS_N]    cold per-chain scalars
    double *TB = CS + 16 * CS_N;                  // [16][NTL][DP] tree vectors: slots 0-7, stack level 1
    double *GB = TB + 16 * NTL * DP;              // [NMAT KS][16][GS] matvec results

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index == chain index in the group
    int cpg = a.cpg > 0 ? a.cpg : 16;  // chains of this workgroup (wave_layout_cpg; the other waves only run matvec jobs)
    int chain = blockIdx.x * cpg + w;
    bool real = w < cpg && chain < a.n_chain;
    if (a.tail_list) {
        // the tail of a launch: the chains that stopped early, listed; as few per workgroup as the CUs allow (the numbers do
        // not depend on the grouping)
        const int cnt = rfl(a.tail_count[0]);
        cpg = cnt <= a.n_cu ? 1 : (cnt <= 2 * a.n_cu ? 2 : 4);
        const int idx = blockIdx.x * cpg + w;
        real = w < cpg && idx < cnt;
        chain = real ? rfl(a.tail_list[idx]) : 0;
    }
    const int d = m.d;
    const bool lane_ok = lane < DP;

    // A operands of this wave's job (matrix, row tile, K part) stay in registers for the whole launch
    double afr[KPJ_P];
    {
        const int slot_m = w / (W * KS_P), rem = w % (W * KS_P), t = rem / KS_P, kp = rem % KS_P;
        const double *Af = (slot_m == 0 ? m.Sf : (slot_m == 1 ? m.Hf : m.Hdf)) + (t * NS + kp * KPJ_P) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KPJ_P; ++s) afr[s] = (w < NJOB_P) ? Af[s * 64] : 0.;
    }
    if (tid < 8) alive[tid] = 0;
    const double c_lin = lane_ok ? m.pd[PD_LIN * DP + lane] : 0.;
    const double c_mu = lane_ok ? m.pd[PD_MU * DP + lane] : 0.;
    const double c_dmu = (DEC == 1 && lane_ok) ? m.pd[PD_DMU * DP + lane] : 0.;
    const double c_smu = lane_ok ? m.pd[PD_SMU * DP + lane] : 0.;
    // constraint transform of this lane's dimension (TR) and what phase A leaves for phase C: x(q), dx/dq,
    // (d2x/dq2) / (dx/dq), log |dx/dq|
    const int c_kind = (TR && lane_ok) ? (int)m.pd[PD_KIND * DP + lane] : 0;
    const double c_lo = (TR && lane_ok) ? m.pd[PD_LO * DP + lane] : 0., c_rg = (TR && lane_ok) ? m.pd[PD_RG * DP + lane] : 1.;
    double xs = 0., jac = 1., gj = 0., logdet_l = 0.;

    // ---- per-chain state (scalars are wave-uniform) ----
    double q = 0., p = 0., g = 0., var = 1.;
    double TLp = 0., TPs = 0., TPq = 0.;  // the subtree under construction: left end p, p_sum, proposal q
    double TRp = 0.;                      // p of the newest accounted leaf (the subtree's right end in time order)
    double L0p = 0., L0q = 0.;            // stack level 0 (a single waiting leaf)
    double TPg = 0., L0g = 0.;            // the gradient at the proposals TPq / L0q (the next iteration may start there)
    uint64_t rs[4] = {0, 0, 0, 0};
    int i_iter = 0, mode = M_DONE, err = 0;
    double eps = 0., eps_t = 0.;
    int dir = 1, depth = 0, i_leaf = 0, n_prop = 0, diverged = 0;
    double start_energy = 0., acc_sum = 0.;
    double T_W = 0., T_acc = 0.;
    double max_de = 0., w_off = 0., L0_W = 0., L0_acc = 0.;
    bool pend = false;                    // a finished leaf evaluation waits for its bookkeeping
    double E_pend = 0., lp_pend = 0.;
    double *csw = CS + w * CS_N;
    auto cs_set = [&](int i, double v) { if (lane == 0) csw[i] = v; };
    auto cs_get = [&](int i) -> double { return rfl(csw[i]); };
    unsigned long long nlf = 0;
    double *sbase = a.scratch + ((size_t)(real ? chain : 0) * a.nslot) * DP + lane;
    double *scp = a.sc + (size_t)(real ? chain : 0) * BFHIP_SC_N;
    double *vecp = a.vec + (size_t)(real ? chain : 0) * BFHIP_VEC_N * d;
    double *lsw = LS + w * (MAXL * LS_N);
    double *tbw = TB + (w * NTL) * DP + lane;
    const int nw = a.cfg.n_warmup;

    auto load_vec = [&](int field, double pad) -> double { return (lane < d) ? vecp[field * d + lane] : pad; };
    auto store_vec = [&](int field, double v) { if (lane < d) vecp[field * d + lane] = v; };
    // tree vectors: slots 0-7 (ends, proposal, p_sum) and stack level 1 in LDS, deeper stack levels in scratch
    auto ldv = [&](int slot) -> double {
        double v = 0.;
        if (lane_ok) {
            if (slot < SL_STACK + 8) v = tbw[(slot < 8 ? slot : slot - 4) * DP];
            else v = sbase[(size_t)slot * DP];
        }
        return v;
    };
    auto stv = [&](int slot, double v) {
        if (lane_ok) {
            if (slot < SL_STACK + 8) tbw[(slot < 8 ? slot : slot - 4) * DP] = v;
            else sbase[(size_t)slot * DP] = v;
        }
    };
    // metric.random: samplers/hmc_utils/metrics.py:83-86 (same stream layout as bf_sampler_kernel)
    auto draw_momentum = [&]() {
        const uint64_t K = bf_xoshiro_next(rs);
        const uint64_t P = (uint64_t)(lane >> 1);
        const double u1 = bf_u01_open0(bf_mix64(K + (2 * P + 1) * BF_GOLDEN));
        const double u2 = bf_u01(bf_mix64(K + (2 * P + 2) * BF_GOLDEN));
        const double rad = sqrt(-2. * log(u1));
        double sn, cs;
        sincospi(2. * u2, &sn, &cs);
        const double z = (lane & 1) ? rad * sn : rad * cs;
        p = (lane < d) ? (1. / sqrt(var)) * z : 0.;
        g = 0.;
    };
    if (real) {
        for (int k = 0; k < 4; ++k) rs[k] = rfl((uint64_t)a.rng[(size_t)chain * 4 + k]);
        cs_set(CS_LOG_STEP, rfl(scp[BFHIP_SC_LOG_STEP]));
        cs_set(CS_LOG_BAR, rfl(scp[BFHIP_SC_LOG_BAR]));
        cs_set(CS_HBAR, rfl(scp[BFHIP_SC_HBAR]));
        cs_set(CS_SMU, rfl(scp[BFHIP_SC_MU]));
        cs_set(CS_COUNT, rfl(scp[BFHIP_SC_COUNT]));
        cs_set(CS_STEP_NOW, uexp(rfl(scp[BFHIP_SC_LOG_STEP])));
        cs_set(CS_STEP_BAR, uexp(rfl(scp[BFHIP_SC_LOG_BAR])));
        i_iter = rfl((int)scp[BFHIP_SC_I_ITER]);
        err = rfl((int)scp[BFHIP_SC_ERROR]);
        q = load_vec(BFHIP_VEC_Q, 0.);
        var = load_vec(BFHIP_VEC_VAR, 1.);
        if (i_iter < a.iter_end && err == 0) {
            mode = M_INIT;
            draw_momentum();
        } else if (a.tail_done && lane == 0) {
            atomicAdd(a.tail_done, 1);   // (nothing left to do in this launch)
        }
    }
    __syncthreads();
    if (lane == 0 && mode == M_DONE) atomicAdd(&alive[6], 1);   // (waves without a chain, chains without work)

#ifdef BF_TRACE
    __shared__ unsigned long long TRC[BF_TRACE * 16];
    for (int i = threadIdx.x; i < BF_TRACE * 16; i += 1024) TRC[i] = 0;
    int trip_no = 0;
#endif
    // ---- BaseHMC.astep start: base_hmc.py:70-76, Tree.__init__: nuts.py:24-43 (state in q, p, g; energy and logp given) ----
    auto init_tree = [&](double E0, double logp0) {
        if (!(fabs(E0) <= 1.7976931348623157e308)) {
            err = 1;
        } else {
            start_energy = E0;
            stv(SL_LEFT_Q, q); stv(SL_LEFT_P, p); stv(SL_LEFT_G, g);
            stv(SL_RIGHT_Q, q); stv(SL_RIGHT_P, p); stv(SL_RIGHT_G, g);
            stv(SL_PROP_Q, q); stv(SL_PSUM, p);
            stv(SL_PROPG, g);
            cs_set(CS_PROP_E, E0);
            cs_set(CS_PROP_LOGP, logp0);
            cs_set(CS_TREE_W, 1.);
            w_off = 0.;
            max_de = 0.;
            depth = 0; acc_sum = 0.; n_prop = 0; diverged = 0; i_leaf = 0;
            // step_size.py:25-29: exp(log_step) while warming up, exp(log_step_bar) after; both are kept up to date for the
            // statistics (CS_STEP_NOW / CS_STEP_BAR), so no exponential is needed here
            eps = (i_iter < nw) ? cs_get(CS_STEP_NOW) : cs_get(CS_STEP_BAR);
            dir = (bf_u01(bf_xoshiro_next(rs)) < 0.5) ? 1 : -1;  // nuts.py:210, log(U) < log(1/2)
            pend = false;
            mode = M_LEAF;
        }
        if (err != 0) mode = M_DONE;
    };
    // ---- iteration end: step-size adaptation, statistics, the new sample, metric adaptation, next momentum ----
    auto iteration_end = [&]() {
        // ================= iteration end (base_hmc.py:80-85) =================
        {
            const bool warm = i_iter < nw;
            const double accept_stat = acc_sum / (double)n_prop;  // nuts.py:186
            double log_step = cs_get(CS_LOG_STEP), log_bar = cs_get(CS_LOG_BAR);
            if (warm && a.cfg.adapt_step_size) {  // step_size.py:31-45
                const double count = cs_get(CS_COUNT);
                const double wgt = 1. / (count + a.cfg.t_0);
                const double hbar = ((1. - wgt) * cs_get(CS_HBAR) + wgt * (a.cfg.target_accept - accept_stat));
                log_step = cs_get(CS_SMU) - hbar * usqrt(count) / a.cfg.gamma;
                const double mk = uexp(-a.cfg.k * ulog(count));  // count ** -k
                log_bar = mk * log_step + (1. - mk) * log_bar;
                cs_set(CS_HBAR, hbar);
                cs_set(CS_LOG_STEP, log_step);
                cs_set(CS_LOG_BAR, log_bar);
                cs_set(CS_COUNT, count + 1.);
                cs_set(CS_STEP_NOW, uexp(log_step));
                cs_set(CS_STEP_BAR, uexp(log_bar));
            }
            const double prop_E = cs_get(CS_PROP_E), prop_logp = cs_get(CS_PROP_LOGP);
            const int orow = i_iter - a.iter_out0;
            if (orow >= 0 && orow < a.n_out) {  // (q is the new sample: phase A of this trip took it from the proposal slot)
                if (lane == 0) {
                    double *st = a.stats + ((size_t)chain * a.n_out + orow) * BFHIP_STAT_STRIDE;
                    st[BFHIP_NS_LOGP] = prop_logp;
                    st[BFHIP_NS_ENERGY] = prop_E;
                    st[BFHIP_NS_TREE_DEPTH] = (double)depth;
                    st[BFHIP_NS_TREE_SIZE] = (double)n_prop;
                    st[BFHIP_NS_MEAN_TREE_ACCEPT] = accept_stat;
                    st[BFHIP_NS_STEP_SIZE] = cs_get(CS_STEP_NOW);
                    st[BFHIP_NS_STEP_SIZE_BAR] = cs_get(CS_STEP_BAR);
                    st[BFHIP_NS_WARMUP] = warm ? 1. : 0.;
                    st[BFHIP_NS_ENERGY_CHANGE] = prop_E - start_energy;
                    st[BFHIP_NS_MAX_ENERGY_CHANGE] = max_de;
                    st[BFHIP_NS_DIVERGING] = (double)diverged;
                }
                double *sp = a.samples + ((size_t)chain * a.n_out + orow) * d;
                if (lane < d) sp[lane] = q;
            }
            // QuadMetricDiagAdapt.update: metrics.py:186-211, _WeightedVariance.add_sample :354-360
            if (warm && a.cfg.adapt_metric) {
                double fg_n = rfl(scp[BFHIP_SC_FG_N]), bg_n = rfl(scp[BFHIP_SC_BG_N]);
                double n_samples = rfl(scp[BFHIP_SC_N_SAMPLES]), prev_upd = rfl(scp[BFHIP_SC_PREV_UPDATE]);
                double adapt_window = rfl(scp[BFHIP_SC_ADAPT_WINDOW]);
                const long delta = (long)(n_samples - prev_upd);
                double fm = load_vec(BFHIP_VEC_FG_MEAN, 0.), fr = load_vec(BFHIP_VEC_FG_RAW, 0.);
                double bm = load_vec(BFHIP_VEC_BG_MEAN, 0.), br = load_vec(BFHIP_VEC_BG_RAW, 0.);
                fg_n += 1.;
                bg_n += 1.;
                double od = q - fm;
                fm += od / fg_n;
                fr += 1. * od * (q - fm);
                od = q - bm;
                bm += od / bg_n;
                br += 1. * od * (q - bm);
                if ((delta + 1) % (long)a.cfg.update_window == 0) {  // metrics.py:181-184
                    if (lane < d) var = fr / fg_n;
                    store_vec(BFHIP_VEC_VAR, var);
                }
                if ((double)delta >= adapt_window) {
                    fm = bm; fr = br; bm = 0.; br = 0.;
                    fg_n = bg_n;
                    bg_n = 10.;
                    prev_upd = n_samples;
                    if (a.cfg.doubling) adapt_window *= 2.;
                }
                n_samples += 1.;
                store_vec(BFHIP_VEC_FG_MEAN, fm);
                store_vec(BFHIP_VEC_FG_RAW, fr);
                store_vec(BFHIP_VEC_BG_MEAN, bm);
                store_vec(BFHIP_VEC_BG_RAW, br);
                if (lane == 0) {
                    scp[BFHIP_SC_FG_N] = fg_n;
                    scp[BFHIP_SC_BG_N] = bg_n;
                    scp[BFHIP_SC_N_SAMPLES] = n_samples;
                    scp[BFHIP_SC_PREV_UPDATE] = prev_upd;
                    scp[BFHIP_SC_ADAPT_WINDOW] = adapt_window;
                }
            }
            i_iter += 1;
            TRACE(12);
            bool go_on = i_iter < a.iter_end && err == 0;
            if (go_on && a.tail_stop > 0 && a.iter_end - i_iter >= 8) {
                // The launch's tail (bfhip_sampler.hip: launch_nuts_pipe): one of the last chains of its workgroup, with at least a
                // quarter of the launch's iterations left while three quarters of the launch's chains are through, stops here
                // and goes on in the launch of the tail (as after any cut between launches).  (The state array still holds the
                // iteration the chain entered the launch with.)
                const int i_iter0 = rfl((int)scp[BFHIP_SC_I_ITER]);
                if (4 * (a.iter_end - i_iter) >= a.iter_end - i_iter0 && 16 - rfl(alive[6]) <= a.tail_stop) {
                    const int dg = rfl(__hip_atomic_load(a.tail_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (4 * dg >= a.tail_q * a.n_chain) go_on = false;
                }
            }
            if (go_on) {
                mode = M_INIT;
                draw_momentum();
            } else {
                mode = M_DONE;
                if (lane == 0) {
                    atomicAdd(&alive[6], 1);   // (chains of this workgroup that have left)
                    if (a.tail_done && (i_iter >= a.iter_end || err != 0)) atomicAdd(a.tail_done, 1);
                }
            }
        }
    };

    auto gb_read = [&](int slot_m) -> double {
        const double *gp = GB + ((slot_m * KS_P) * 16 + w) * GS + lane;
        double r = gp[0];
        if (KS_P > 1) r += gp[16 * GS];
        return r;
    };

    for (int trip = 0;; ++trip) {
#ifdef BF_TRACE
        trip_no = trip;
#endif
        TRACE(0);
        // ================= phase A: first half of the (speculative) leapfrog step, B operands =================
        bool evaluating = false;
        if (mode == M_INIT) {
            evaluating = true;  // compute_state at the start of a launch (base_hmc.py:70): a step of length 0
            eps_t = 0.;
        } else if (mode == M_LEAF) {
            int dir_use = dir;
            evaluating = true;
            if (pend && i_leaf == (1 << depth) - 1) {
                // the leaf in flight closes its doubling: park the new end (its p waits in TRp until the pending
                // full-tree checks have read the p of the end it replaces) and start the next doubling in the direction
                // the stream will give it
                const int eo = (dir > 0) ? SL_RIGHT_Q : SL_LEFT_Q;
                stv(eo + 0, q); stv(eo + 2, g);
                if (depth + 1 >= a.cfg.max_treedepth) {
                    evaluating = false;  // the tree stops at this depth whatever the checks say
                } else {
                    uint64_t t[4] = {rs[0], rs[1], rs[2], rs[3]};
                    for (int k = 0; k <= depth; ++k) (void)bf_xoshiro_next(t);  // `depth` merges and the swap
                    dir_use = (bf_u01(bf_xoshiro_next(t)) < 0.5) ? 1 : -1;
                    if (dir_use != dir) {
                        const int eo2 = (dir_use > 0) ? SL_RIGHT_Q : SL_LEFT_Q;
                        q = ldv(eo2 + 0); p = ldv(eo2 + 1); g = ldv(eo2 + 2);
                    }
                }
            }
            if (evaluating) eps_t = eps * (double)dir_use;
        }
        if (evaluating) {
            {
                const double dt = 0.5 * eps_t;
                p = p + dt * g;                    // integration.py:80
                q = q + eps_t * (var * p);         // :82-85
            }
            xs = q;
            if constexpr (TR) {
                double J, J2;
                bf_to_original(q, c_kind, c_lo, c_rg, xs, J, J2);
                logdet_l = 0. + log(fabs(J));
                jac = J;
                gj = J2 / J;
            }
            if (lane_ok) {
                const int xi = (lane >> 2) * XS + w + 16 * (lane & 3);  // B[k = dim&3][n = chain] of k-step dim>>2
                XB[xi] = xs;
                XB[NS * XS + xi] = xs - c_mu;
                if constexpr (DEC == 1) XB[2 * NS * XS + xi] = xs - c_dmu;   // (the decay term lives in the original space)
            }
        }
        if (lane == 0) {
            if (mode != M_DONE) alive[trip & 1] = 1;
            if (evaluating) atomicOr((unsigned *)&alive[2 + (trip & 1)], 1u << w);
        }
        TRACE(1);
        __syncthreads();  // B1
        TRACE(2);
        if (rfl(alive[trip & 1]) == 0) break;  // every chain of the group is done (uniform)
        const unsigned ev_mask = (unsigned)rfl(alive[2 + (trip & 1)]);
        if (tid == 0) { alive[(trip + 1) & 1] = 0; alive[2 + ((trip + 1) & 1)] = 0; alive[4 + ((trip + 1) & 1)] = 0; }

        // ================= phase B: gradient tiles on MFMA, the pending bookkeeping between them =================
        // A wave's MFMAs form one dependent chain (the accumulator); a dependent v_mfma_f64_16x16x4_f64 issues about
        // 256 cycles after its predecessor while the pipe takes a new one every 64 cycles, from four waves in turn.
        // The chain is spread over the stages of the pending bookkeeping, one MFMA at each of KPJ fixed points of the
        // code below, so that a wave waiting for its accumulator leaves the issue slots to the stages of the others.
        // (Stages that have nothing to do fall through; the MFMAs then simply queue up.  The FP64 VALU work does not
        // overlap with the MFMAs themselves: they share the pipe.)
        const bool job = ev_mask != 0 && w < NJOB_P;
        typename PipeAcc<QUAD>::type acc = {};
        const double *Xf = XB + ((w / (W * KS_P)) * NS + (w % KS_P) * KPJ_P) * XS + (QUAD ? (lane & ~15) + (lane & 3) : lane);
        double x_pre[MPS], x_pr2[QUAD == 2 ? MPS : 1];
#pragma unroll
        for (int u = 0; u < MPS; ++u) x_pre[u] = job ? Xf[u * XS] : 0.;
        if constexpr (QUAD == 2) {
#pragma unroll
            for (int u = 0; u < MPS; ++u) x_pr2[u] = job ? Xf[u * XS + 4] : 0.;
        }
#define BF_MF(K)                                                                                            \
        do {                                                                                                \
            if ((K) * MPS < KPJ_P) {                                                                        \
                double x_cur[MPS], x_cu2[QUAD == 2 ? MPS : 1];                                             \
                _Pragma("unroll") for (int u = 0; u < MPS; ++u) x_cur[u] = x_pre[u];                        \
                if constexpr (QUAD == 2) { _Pragma("unroll") for (int u = 0; u < MPS; ++u) x_cu2[u] = x_pr2[u]; } \
                if (((K) + 1) * MPS < KPJ_P && job) {                                                       \
                    _Pragma("unroll") for (int u = 0; u < MPS; ++u) x_pre[u] = Xf[((((K) + 1) * MPS < KPJ_P ? ((K) + 1) * MPS : 0) + u) * XS]; \
                    if constexpr (QUAD == 2) { _Pragma("unroll") for (int u = 0; u < MPS; ++u) x_pr2[u] = Xf[((((K) + 1) * MPS < KPJ_P ? ((K) + 1) * MPS : 0) + u) * XS + 4]; } \
                }                                                                                           \
                _Pragma("unroll") for (int u = 0; u < MPS; ++u) {                                           \
                    asm volatile("" : "+v"(acc) : : "memory");                                              \
                    if constexpr (QUAD == 2) {                                                              \
                        if (job) {                                                                          \
                            acc[0] = bf_pipe_mfma(afr[((K) * MPS < KPJ_P ? (K) * MPS : 0) + u], x_cur[u], acc[0]); \
                            acc[1] = bf_pipe_mfma(afr[((K) * MPS < KPJ_P ? (K) * MPS : 0) + u], x_cu2[u], acc[1]); \
                        }                                                                                   \
                    } else {                                                                                \
                        if (job) acc = bf_pipe_mfma(afr[((K) * MPS < KPJ_P ? (K) * MPS : 0) + u], x_cur[u], acc); \
                    }                                                                                       \
                    asm volatile("" : "+v"(acc) : : "memory");                                              \
                }                                                                                           \
            }                                                                                               \
        } while (0)
        int unit = pend ? U_EVAL : U_DONE, lev = 0;
        bool ended = false;
        pend = false;
        BF_MF(0);
        TRACE(3);
        double dE = 0.;
        if (unit == U_EVAL) {
            // ---- Tree._single_step: nuts.py:105-132 ----
            nlf += 1;
            n_prop += 1;
            dE = E_pend - start_energy;
            if (dE != dE) dE = INFINITY;
            if (fabs(dE) > fabs(max_de)) max_de = dE;
            cs_set(CS_T_E, E_pend);
            cs_set(CS_T_LOGP, lp_pend);
            T_acc = 0.;
            if (!(fabs(dE) < a.cfg.max_change)) {
                diverged = 1;
                unit = U_ABORT;
            }
        }
        BF_MF(1);
        if (unit == U_EVAL) {
            // multinomial weight exp(-dE) relative to a running offset w_off (exact streaming log-sum-exp)
            double aw = -dE - w_off;
            if (aw > 600.) {
                const double sc_ = uexp(-aw);
                cs_set(CS_TREE_W, cs_get(CS_TREE_W) * sc_);
                if (lane == 0)
                    for (int l2 = 0; l2 < depth; ++l2) lsw[l2 * LS_N + LS_LS] *= sc_;
                L0_W *= sc_;
                w_off = w_off + aw;
                aw = 0.;
            }
            T_W = uexp(aw);
            const double pacc = (w_off == 0.) ? T_W : uexp(-dE);
            T_acc = pacc > 1. ? 1. : pacc;
            TLp = TRp;
            TPs = TRp;  // (TPq was set when the evaluation finished)
            unit = U_MERGE;
        }
        BF_MF(2);
        TRACE(9);
        if (unit == U_MERGE && (i_leaf & 1) && depth > 0) {
            // ---- level-0 merge with the previous leaf, whose (p, q) wait in L0p / L0q (nuts.py:146-178) ----
            const double ps0 = L0p + TRp;
            double r2[2] = {ps0 * (var * L0p), ps0 * (var * TRp)};  // nuts.py:150-151
            wave_sum_n<2>(r2);
            T_acc = L0_acc + T_acc;  // :173
            const double Wsum = L0_W + T_W;
            if (Wsum != Wsum) err = 2;
            const double u = bf_u01(bf_xoshiro_next(rs));  // :163-167, drawn even when turning
            lev = 1;
            if ((r2[0] <= 0.) || (r2[1] <= 0.)) {
                unit = U_ABORT;
            } else {
                if (!((u * Wsum < T_W) || (u == 0.))) {
                    TPq = L0q;
                    TPg = L0g;
                    cs_set(CS_T_E, rfl(lsw[LS_E]));
                    cs_set(CS_T_LOGP, rfl(lsw[LS_LOGP]));
                }
                T_W = Wsum;
                TPs = L0p + TRp;
                TLp = L0p;
            }
        }
        BF_MF(3);
        // ---- merge upwards while the finished subtree is a right child (nuts.py:146-178) ----
        while (unit == U_MERGE && lev < depth && ((i_leaf >> lev) & 1)) {
            const int slot = SL_STACK + 4 * lev;
            const double A = ldv(slot + 0), B = ldv(slot + 1), S1 = ldv(slot + 2);
            const double psum = S1 + TPs;
            const double vA = var * A, vB = var * B, vC = var * TLp, vD = var * TRp;
            const double ps1 = S1 + TLp;   // :155-157
            const double ps2 = B + TPs;    // :158-160
            double r6[6] = {psum * vA, psum * vD, ps1 * vA, ps1 * vC, ps2 * vB, ps2 * vD};
            wave_sum_n<6>(r6);
            const bool turning = (r6[0] <= 0.) || (r6[1] <= 0.) || (r6[2] <= 0.) || (r6[3] <= 0.) || (r6[4] <= 0.) || (r6[5] <= 0.);
            const double *lsp = lsw + lev * LS_N;
            T_acc = rfl(lsp[LS_ACC]) + T_acc;  // :173
            const double Wsum = rfl(lsp[LS_LS]) + T_W;
            if (Wsum != Wsum) err = 2;
            const double u = bf_u01(bf_xoshiro_next(rs));  // consumed even when this merge's check says turning
            const bool keep_t2 = (u * Wsum < T_W) || (u == 0.);
            lev += 1;
            if (turning) {
                unit = U_ABORT;  // ancestors above this level still add their accept sums
            } else {
                if (!keep_t2) {
                    TPq = ldv(slot + 3);  // the sibling's proposal
                    TPg = ldv(SL_PG + lev - 1);
                    cs_set(CS_T_E, rfl(lsp[LS_E]));
                    cs_set(CS_T_LOGP, rfl(lsp[LS_LOGP]));
                }
                T_W = Wsum;
                TLp = A;
                TPs = psum;
            }
        }
        BF_MF(4);
        TRACE(10);
        if (unit == U_MERGE) {
            if (lev < depth) {
                // the subtree waits for its right sibling
                if (lev == 0) {
                    L0p = TRp;
                    L0q = TPq;  // a single leaf: its proposal is its own position
                    L0g = TPg;
                    L0_W = T_W;
                    L0_acc = T_acc;
                } else {
                    const int slot = SL_STACK + 4 * lev;
                    stv(slot + 0, TLp); stv(slot + 1, TRp); stv(slot + 2, TPs); stv(slot + 3, TPq);
                    stv(SL_PG + lev, TPg);
                }
                if (lane == 0) {
                    double *lsp = lsw + lev * LS_N;
                    lsp[LS_LS] = T_W; lsp[LS_ACC] = T_acc;
                    lsp[LS_E] = (lev == 0) ? E_pend : cs_get(CS_T_E);
                    lsp[LS_LOGP] = (lev == 0) ? lp_pend : cs_get(CS_T_LOGP);
                }
                i_leaf += 1;
                unit = U_DONE;
            } else {
                unit = U_DBL_END;
            }
        }
        BF_MF(5);
        if (__builtin_expect(unit == U_ABORT, 0)) {
            // unwind: every pending ancestor adds its left half's accept_sum (nuts.py:173)
            for (int al = (diverged ? 0 : lev); al < depth; ++al)
                if ((i_leaf >> al) & 1) T_acc = rfl(lsw[al * LS_N + LS_ACC]) + T_acc;
            depth += 1;  // nuts.py:71-73
            acc_sum += T_acc;
            unit = U_END1;
        } else if (unit == U_DBL_END) {
            // ---- Tree.extend after a complete subtree: nuts.py:71-103 ----
            // (phase A has already parked the q and the gradient of the new end; its p is still TRp)
            double ps = ldv(SL_PSUM);
            const double oldL = ldv(SL_LEFT_P), oldR = ldv(SL_RIGHT_P);
            depth += 1;
            acc_sum += T_acc;
            {   // :81-83  logbern(ls_new - ls_old)  <=>  U * W_old < W_new
                const double tree_W = cs_get(CS_TREE_W);
                if (T_W != T_W || tree_W != tree_W) err = 2;
                const double u = bf_u01(bf_xoshiro_next(rs));
                if ((u * tree_W < T_W) || (u == 0.)) {
                    stv(SL_PROP_Q, TPq);
                    stv(SL_PROPG, TPg);
                    cs_set(CS_PROP_E, cs_get(CS_T_E));
                    cs_set(CS_PROP_LOGP, cs_get(CS_T_LOGP));
                }
                cs_set(CS_TREE_W, tree_W + T_W);  // :85
            }
            ps += TPs;  // :86 (in place)
            const double vN = var * TRp, vT = var * TLp, vL = var * oldL, vR = var * oldR;
            double r6[6];
            // NOTE (reference behaviour, kept on purpose): leftmost_p_sum (dir > 0) / rightmost_p_sum (dir < 0) alias
            // self.p_sum, which line 86 has just updated in place.
            if (dir > 0) {
                const double ps1 = ps + TLp, ps2 = oldR + TPs;
                r6[0] = ps * vL; r6[1] = ps * vN; r6[2] = ps1 * vL; r6[3] = ps1 * vT; r6[4] = ps2 * vR; r6[5] = ps2 * vN;
            } else {
                const double ps1 = TPs + oldL, ps2 = TLp + ps;
                r6[0] = ps * vN; r6[1] = ps * vR; r6[2] = ps1 * vN; r6[3] = ps1 * vL; r6[4] = ps2 * vT; r6[5] = ps2 * vR;
            }
            wave_sum_n<6>(r6);
            stv(SL_PSUM, ps);
            stv((dir > 0) ? SL_RIGHT_P : SL_LEFT_P, TRp);
            const bool turning = (r6[0] <= 0.) || (r6[1] <= 0.) || (r6[2] <= 0.) || (r6[3] <= 0.) || (r6[4] <= 0.) || (r6[5] <= 0.);
            if (turning || depth >= a.cfg.max_treedepth) {
                unit = U_END1;
            } else {
                dir = (bf_u01(bf_xoshiro_next(rs)) < 0.5) ? 1 : -1;  // nuts.py:210 (phase A read this draw ahead)
                i_leaf = 0;
                unit = U_DONE;
            }
        }
        BF_MF(6);
        TRACE(11);
        if (__builtin_expect(unit == U_END1, 0)) {   // (cold: once per tree; the register allocator keeps its spill code here)
            ended = true;  // the tree the evaluation in flight belongs to has ended: the evaluation is dropped
            if (err == 0) {
                // the proposal becomes the sample and the start of the next iteration.  base_hmc.py:70 evaluates it again
                // (compute_state); value and gradient are the ones of the leaf it was (kept next to its position through
                // the merges), so the iteration starts right here, without that evaluation
                const double g_prop = ldv(SL_PROPG);  // (issued early: global scratch)
                q = ldv(SL_PROP_Q);
                iteration_end();  // -> M_INIT with a fresh momentum, or M_DONE
                if (mode == M_INIT) {
                    g = g_prop;
                    const double logp0 = cs_get(CS_PROP_LOGP);
                    const double kin0 = wave_sum(p * (var * p));   // metrics.py:88-91
                    init_tree(0.5 * kin0 - logp0, logp0);          // integration.py:28-34
                }
            }
        }
        if (err != 0) { mode = M_DONE; ended = true; }
        BF_MF(7);
#undef BF_MF
        TRACE(4);
        if (job) {
            const int mc = lane & 15, mg = lane >> 4;
            const int slot_m = w / (W * KS_P), rem = w % (W * KS_P), t = rem / KS_P, kp = rem % KS_P;
            if constexpr (QUAD == 1) {
                GB[((slot_m * KS_P + kp) * 16 + (lane & 3)) * GS + 16 * t + 4 * ((lane >> 2) & 3) + (lane >> 4)] = acc;
            } else if constexpr (QUAD == 2) {
                GB[((slot_m * KS_P + kp) * 16 + (lane & 3)) * GS + 16 * t + 4 * ((lane >> 2) & 3) + (lane >> 4)] = acc[0];
                GB[((slot_m * KS_P + kp) * 16 + 4 + (lane & 3)) * GS + 16 * t + 4 * ((lane >> 2) & 3) + (lane >> 4)] = acc[1];
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) GB[((slot_m * KS_P + kp) * 16 + mc) * GS + 16 * t + 4 * r4 + mg] = acc[r4];
            }
        }
        TRACE(5);
        __syncthreads();  // B2
        TRACE(6);

        // ================= phase C: finish the evaluation =================
        // what follows a complete evaluation (f, gn: the surrogate's value and gradient in its own coordinates)
        auto finish = [&](double f, double gn, bool kin_ready, double r_kin, double logdet, double r_bd2, double dgr) {
            gn = gn * jac;  // chain rule (module.py:226, density.py:558); 1 without the transform
            if constexpr (DEC) {  // density.py:740-746
                f -= m.decay_gamma * bf_clip0(r_bd2 - m.decay_alpha2);
                if (r_bd2 > m.decay_alpha2) gn -= 2. * m.decay_gamma * dgr;
            }
            if constexpr (TR) {  // density.py:747-750
                f += logdet;
                gn += gj;
            }
            const double logp_new = f;
            // second half of the leapfrog and the kinetic energy
            const double dt = 0.5 * eps_t;
            p = p + dt * gn;        // integration.py:90
            g = gn;
            double kin = p * (var * p);   // metrics.py:88-91
            kin = kin_ready ? r_kin : wave_sum(kin);
            const double E_new = 0.5 * kin - logp_new;  // integration.py:92-93
            if (mode == M_INIT) {
                init_tree(E_new, logp_new);
            } else {
                pend = true;
                E_pend = E_new;
                lp_pend = logp_new;
                TRp = p;
                TPq = q;
                TPg = g;
            }
        };
        // Outside the bound (modules/poly.py:480-503) the surrogate is wanted at the projected point x_0 = mu + t (x - mu),
        // t = alpha / beta.  It is linear + quadratic, so S x_0 = S mu + t (S x - S mu) follows from the S x of THIS trip and
        // the per-dimension table's S mu -- no second pass over the tiles, no trip of its own (M_OOB is not used by this
        // kernel) -- and the two sums of the reference's formulas are polynomials in t of two t-free sums:
        //   a1 = (x - mu) . (S mu + lin),  a2 = (x - mu) . S (x - mu)   (bfhip_oob.h).
        if (evaluating && !ended) {
            const double sx = lane_ok ? gb_read(0) : 0.;
            const double hv = lane_ok ? gb_read(1) : 0.;
            const double dgr = DEC == 2 ? hv : ((DEC == 1 && lane_ok) ? gb_read(2) : 0.);   // H_decay (x - mu_decay); DEC = 2: the bound's product
            double gn = sx + c_lin;
            const double xm = xs - c_mu;
            constexpr bool fast_kin = !DEC;   // (with the decay term the gradient is final only after its sum)
            // (the surrogate's value, linear + quadratic term, summed per lane: one reduction for both)
            const double sv = sx - c_smu, gmu = c_smu + c_lin;   // S (x - mu), the gradient at mu (bfhip_oob.h)
            double r_kin = 0., r_val, r_b2, r_bd2 = 0., r_a[2] = {0., 0.};
            TRACE(7);
            if constexpr (DEC == 1) {
                // (the densities that carry the decay term live outside the bound: its sum and the two sums of the extrapolation
                // ride in the first reduction -- the same numbers as reductions of their own)
                double r5[5] = {(xs - c_dmu) * dgr, __builtin_fma(0.5 * xs, sx, c_lin * xs), xm * hv, xm * gmu, xm * sv};
                wave_sum_n<5>(r5);
                r_bd2 = r5[0]; r_val = r5[1]; r_b2 = r5[2]; r_a[0] = r5[3]; r_a[1] = r5[4];
            } else if constexpr (DEC == 2) {
                // (the decay term's radius IS the bound's: (x - mu_d) . H_d (x - mu_d) = (x - mu) . H (x - mu), the same products summed
                // the same way)
                double r4[4] = {__builtin_fma(0.5 * xs, sx, c_lin * xs), xm * hv, xm * gmu, xm * sv};
                wave_sum_n<4>(r4);
                r_val = r4[0]; r_b2 = r4[1]; r_a[0] = r4[2]; r_a[1] = r4[3];
                r_bd2 = r_b2;
            } else {
                double r3[3] = {0., __builtin_fma(0.5 * xs, sx, c_lin * xs), xm * hv};
                {  // in-bound gradient is already final: the kinetic energy rides along
                    double ge = gn * jac;
                    if constexpr (TR) ge += gj;
                    const double pe = p + (0.5 * eps_t) * ge;
                    r3[0] = pe * (var * pe);
                }
                wave_sum_n<3>(r3);
                r_kin = r3[0]; r_val = r3[1]; r_b2 = r3[2];
            }
            TRACE(8);
            double logdet = 0.;
            if constexpr (TR) logdet = wave_sum(logdet_l);
            double f = (m.c0 + r_val) + 0.;
            double beta = 0.;
            const double a2 = m.alpha * m.alpha;
            if (!(r_b2 < a2 * (1. - 1e-12))) beta = usqrt(r_b2);
            bool kin_ready = fast_kin;
            if (beta > m.alpha) {
                if constexpr (!DEC) {
                    r_a[0] = xm * gmu; r_a[1] = xm * sv;
                    wave_sum_n<2>(r_a);
                }
                const BfOob o = bf_oob_scalars(m.alpha, m.inv_alpha, m.f_mu, m.f_poly_mu, beta, r_a[0], r_a[1]);
                f = o.f;
                gn = bf_oob_grad(o, gmu, sv, hv);
                kin_ready = false;
            }
            finish(f, gn, kin_ready, r_kin, logdet, r_bd2, dgr);
        }
    }

#ifdef BF_TRACE
    if (a.stamps && w == 0 && blockIdx.x == 0)
        for (int i = lane; i < BF_TRACE * 16; i += 64) a.stamps[i] = TRC[i];
#endif
    // ---- write the chain state back ----
    if (real) {
        store_vec(BFHIP_VEC_Q, q);
        if (lane == 0) {
            for (int k = 0; k < 4; ++k) a.rng[(size_t)chain * 4 + k] = rs[k];
            scp[BFHIP_SC_LOG_STEP] = cs_get(CS_LOG_STEP);
            scp[BFHIP_SC_LOG_BAR] = cs_get(CS_LOG_BAR);
            scp[BFHIP_SC_HBAR] = cs_get(CS_HBAR);
            scp[BFHIP_SC_COUNT] = cs_get(CS_COUNT);
            scp[BFHIP_SC_I_ITER] = (double)i_iter;
            scp[BFHIP_SC_ERROR] = (double)err;
            if (a.n_leapfrog && nlf) atomicAdd(a.n_leapfrog, nlf);
        }
    }
}
