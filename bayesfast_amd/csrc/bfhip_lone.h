// bfhip_lone.h -- latency-first NUTS for lone and few chains on the common surrogate at d <= 64 (gfx950).
// Included by bfhip_sampler.hip after bfhip_nuts_pipe.h (uses its geometry, enums and helpers).
//
// What it is for.  A launch of the wave-per-chain kernels lasts as long as its busiest chain, and a chain's leapfrog steps are
// sequential: at the end of a launch with stragglers (BASELINE config 3, first round: a few chains build 1023-leaf trees at
// every iteration), and in launches with fewer chains than the chip has room for (config 2: 1024 chains), the rate is
// 1 / (latency of ONE chain's trip).  In bf_nuts_pipe_kernel that trip is 4.7-6.8 k cycles for a chain alone in its
// workgroup (profiles/r05a_trace_plain256.log): phase A 0.6 k, the bookkeeping of the previous leaf 2.9 k -- run by the chain's
// own wave between the two barriers --, phase C 0.9 k, and only 0.5 k of matrix instructions.  FP64 vector instructions have a
// dependent-issue latency of 32 cycles on gfx950 (tools/probe/latency_probe.hip), a wave reduction 144, exp 140, log 470; a
// workgroup barrier costs 45-60.  So the remedy is not fewer instructions but fewer of them in a row:
//
//   * ONE CHAIN PER WORKGROUP, three roles on 2 + W waves.
//     - wave 0, the INTEGRATOR: the leapfrog step and the end of the evaluation (phases A and C of the pipelined kernel,
//       the same expressions in the same order) and nothing else.
//     - wave 1, the BOOKKEEPER: the NUTS tree, one leaf behind -- Tree._single_step, the merges with their U-turn sums and
//       multinomial draws, Tree.extend, the iteration's end (step size, statistics, sample, metric window, next momentum);
//       all of the chain's scalars in ITS registers (no parking in LDS), the whole subtree stack in LDS (no global scratch).
//     - waves 2 .. 1 + W, the MATVEC waves: wave 2 + t owns row tile t of S, H (and the decay term's matrix): NMAT x KS
//       independent accumulation chains of v_mfma_f64_4x4x4_4b per wave, A fragments in registers for the whole launch, the
//       B operand read as 128-byte rows of a transposed copy of x.
//   * Three barriers per trip (B0, B1, B2).  The integrator computes leaf t between B0(t) and B0(t + 1) while the
//     bookkeeper accounts for leaf t - 1; at B0 the bookkeeper's VERDICT becomes visible: go on (and where the doubling in
//     flight ends, with the direction of the next one read ahead from the random stream, as in the pipelined kernel), start
//     a new tree from the proposal (position, cached gradient, fresh momentum, step size), or stop.  The speculation is the
//     pipelined kernel's: one leaf, dropped when the tree ends.
//   * The arithmetic per chain is the pipelined kernel's, sum for sum: the tiles accumulate the same K parts in the same
//     order (v_mfma_f64_4x4x4_4b = the 16 x 16 x 4 tile's sequential chain per entry), the reductions are wave_sum_n, the
//     bookkeeping is the same code on the same values.  Samples, statistics, adapted state and random streams are
//     bit-identical to bf_nuts_pipe_kernel (tests/test_gpu_sampler.py::test_lone_kernel_is_bit_identical_to_pipelined_kernel).
//
// References: samplers/hmc_utils/integration.py:68-95 (leapfrog), samplers/nuts.py:105-178 (tree), :45-103 (extend),
// samplers/hmc_utils/base_hmc.py:62-85 (astep), step_size.py:31-45, metrics.py:186-211.

// inline libm here (the translation unit's exp / log / sqrt / sincospi are macros for out-of-line wrappers of the same
// functions: a call costs the wrapper's prologue and a round of register moves, which is what this kernel exists to avoid)
#pragma push_macro("exp")
#pragma push_macro("log")
#pragma push_macro("sqrt")
#pragma push_macro("sincospi")
#undef exp
#undef log
#undef sqrt
#undef sincospi
// (the out-of-line wrappers round their argument and their result as the call boundary does; inlined, the library's first and
// last operations could contract with the caller's -- step sizes one ulp off the pipelined kernel's were the symptom -- so
// arguments and results pass through an opaque register move)
__device__ inline double ln_opq(double x) { asm("" : "+v"(x)); return x; }
#define LN_EXPV(x) ln_opq(exp(ln_opq(x)))
#define LN_LOGV(x) ln_opq(log(ln_opq(x)))
#define LN_SQRTV(x) ln_opq(sqrt(ln_opq(x)))
#define LN_EXP(x) rfl(LN_EXPV(x))
#define LN_LOG(x) rfl(LN_LOGV(x))
#define LN_SQRT(x) rfl(LN_SQRTV(x))

enum { LN_CONT = 0, LN_NEW = 1, LN_DONE = 2 };
enum { LF_NONE = 0, LF_LEAF = 1, LF_INIT = 2 };
// verdict words (int): command | the leaf in flight closes its doubling | slot of the end it becomes | go on evaluating |
// direction (NEW: of the first doubling; closing: of the next one) | NEW: the evaluation that opens a launch (step of length 0)
enum { LV_CMD = 0, LV_CLOSE, LV_EO, LV_EVAL, LV_DIR, LV_INIT, LV_N = 8 };

template <int W, int DEC>
struct LoneGeo {
    using PG = PipeGeo<W, DEC>;
    static constexpr int DP = 16 * W, NS = 4 * W, NMAT = PG::NMAT, KS = PG::KS, KPJ = PG::KPJ;
    static constexpr int NW = 1 + W, NJ = NMAT * KS;      // waves: integrator (+ row tile 0), bookkeeper, row tiles 1 .. W - 1
    static constexpr int NSLOT = SL_PIPE_N;
    static constexpr int o_XT = 0;                          // [NMAT][4][NS]  x, x - mu, x - mu_decay as B-operand rows
    static constexpr int o_GB = o_XT + NMAT * DP;           // [NMAT KS][DP]  partial products
    static constexpr int o_LF = o_GB + NMAT * KS * DP;      // [3][DP] + 4    the finished leaf: q, p, g | E, logp
    static constexpr int o_NI = o_LF + 3 * DP + 4;          // [4][DP]        a new tree's start: q, p, g, var
    static constexpr int o_VD = o_NI + 4 * DP;              // [2] doubles (eps) + LV_N ints
    static constexpr int o_WF = o_VD + 2 + LV_N / 2;        // [4][DP] + 6    the metric's Welford windows (bookkeeper only)
    static constexpr int o_LS = o_WF + 4 * DP + 6;          // [MAXL][LS_N]   stack scalars
    static constexpr int o_TB = o_LS + BFHIP_MAX_TREEDEPTH * LS_N;   // [NSLOT][DP] tree vectors, every stack level
    static constexpr int n_doubles = o_TB + NSLOT * DP;
};

#ifdef BF_LTRACE
#ifndef BF_LTRACE_IDS   // which stamps are compiled in (a stamp costs ~100 cycles: s_memtime and the wait for it)
#define BF_LTRACE_IDS 0xffff
#endif
#define LTRACE(role, k) do { if (((BF_LTRACE_IDS >> (k)) & 1) && blockIdx.x == 0 && ltrip < BF_LTRACE && lane == 0) LTRC[(ltrip * 2 + (role)) * 16 + (k)] = clock64(); } while (0)
#else
#define LTRACE(role, k) do { } while (0)
#endif

// The matvec jobs of a trip on v_mfma_f64_4x4x4_4b.  A job is (matrix, K part, row tile): one accumulation chain of KPJ k-steps,
// the pipelined kernel's.  A fragment of k-step s: lane 16 k + 4 b + i holds M[16 t + 4 b + i][4 s + k] (the 16 x 16 x 4 tile's
// fragment); B: the lane's k = lane >> 4, every column the chain's x (only column 0 is read back); D: lane 16 i + 4 b + j = row
// 4 b + i, column j.  The NJT = NMAT x KS x W jobs are dealt over NT waves (wave r: jobs r, r + NT, ...), each wave running its
// jobs as independent chains side by side, A fragments in registers for the whole launch.
template <int W, int DEC, int NT>
struct LoneJobs {
    using LG = LoneGeo<W, DEC>;
    static constexpr int NJT = LG::NMAT * LG::KS * W, JPW = (NJT + NT - 1) / NT;
    double afr[JPW][LG::KPJ];
    int xoff[JPW], goff[JPW];
    __device__ inline void load(const DevModel &m, int r, int lane) {
#pragma unroll
        for (int u = 0; u < JPW; ++u) {
            const int j = r + u * NT, jj = j < NJT ? j : 0;
            const int mat = jj / (LG::KS * W), rem = jj % (LG::KS * W), kp = rem / W, t = rem % W;
            const double *Af = (mat == 0 ? m.Sf : (mat == 1 ? m.Hf : m.Hdf)) + (t * LG::NS + kp * LG::KPJ) * 64 + lane;
#pragma unroll
            for (int s = 0; s < LG::KPJ; ++s) afr[u][s] = Af[s * 64];
            xoff[u] = (mat * 4 + (lane >> 4)) * LG::NS + kp * LG::KPJ;
            goff[u] = (j < NJT && (lane & 3) == 0) ? (mat * LG::KS + kp) * LG::DP + 16 * t + 4 * ((lane >> 2) & 3) + (lane >> 4) : -1;
        }
    }
    __device__ inline void run(const double *XT, double *GB) const {
        double xb[JPW][LG::KPJ];
#pragma unroll
        for (int u = 0; u < JPW; ++u) {
            const double *xp = XT + xoff[u];
#pragma unroll
            for (int s = 0; s < LG::KPJ; ++s) xb[u][s] = xp[s];
        }
        double acc[JPW];
#pragma unroll
        for (int u = 0; u < JPW; ++u) acc[u] = 0.;
#pragma unroll
        for (int s = 0; s < LG::KPJ; ++s) {
#pragma unroll
            for (int u = 0; u < JPW; ++u) acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(afr[u][s], xb[u][s], acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < JPW; ++u)
            if (goff[u] >= 0) GB[goff[u]] = acc[u];
    }
};

// Waves of the workgroup: W job waves -- wave 0 is also the integrator, which waits for the products anyway -- and the bookkeeper,
// wave W.  (Waves are dealt to the four SIMDs in turn, and FP64 vector instructions share a SIMD's pipe with its matrix
// instructions.  Measured at d = 64, profiles/r05_lone_layouts.log: four job waves on four SIMDs with the bookkeeper next to the
// integrator, 1.76 us per leapfrog step; three job waves on SIMDs 1-3 and a SIMD without jobs for integrator and bookkeeper, 1.87:
// the bookkeeper's exponential runs 500 cycles shorter and the jobs 400 longer.)
// (The same with the decay term, whose jobs are twelve chains of sixteen k-steps: three job waves of four chains, 1.72 against 1.67 us.)
// FORM 1 (d > 32 only): three job waves instead of four -- a workgroup of four waves, two of them a CU at 256 registers a wave, for
// launches of up to two chains per CU (the jobs take 48 matrix instructions per wave instead of 32).
template <int W, int DEC, int FORM = 0> struct LoneWaves {
    static constexpr bool ITILE = true;                   // the integrator runs jobs
    // waves that run jobs.  d <= 32: the integrator alone -- its 8 chains of 4 k-steps cost 250 cycles more than shared with a second
    // wave, but a workgroup is then two waves and four of them fit a CU at 256 registers a wave, without the bookkeeper's spills:
    // 32-d x 1024 chains 4.18 -> 4.47 x 10^8 (profiles/r05_lone_layouts.log)
    static constexpr int NT = W <= 2 ? 1 : (FORM == 1 ? 3 : W);
    static constexpr int KW = ITILE ? NT : NT + 1;        // the bookkeeper's wave
    static constexpr int NW = KW + 1;
};

template <int W, bool TR, int DEC, int MINW, int FORM = 0>   // DEC: 0 none, 1 the decay term's own matrix, 2 the bound's (bfhip_nuts_pipe.h)
__global__ __launch_bounds__((LoneWaves<W, DEC, FORM>::NW * 64), MINW) void bf_lone_kernel(DevModel m, SamplerArgs a) {
    using LG = LoneGeo<W, DEC>;
    using LWV = LoneWaves<W, DEC, FORM>;
    constexpr int DP = LG::DP, NS = LG::NS, KS = LG::KS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *XT = lds + LG::o_XT, *GB = lds + LG::o_GB, *LF = lds + LG::o_LF, *NI = lds + LG::o_NI, *VD = lds + LG::o_VD;
    int *VI = (int *)(VD + 2);
    double *LS = lds + LG::o_LS, *TB = lds + LG::o_TB;
#ifdef BF_LTRACE
    __shared__ unsigned long long LTRC[BF_LTRACE * 32];
    for (int i = threadIdx.x; i < BF_LTRACE * 32; i += LWV::NW * 64) LTRC[i] = 0;
    int ltrip = 0;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int chain = blockIdx.x;
    if (a.tail_list) {
        const int cnt = rfl(a.tail_count[0]);
        if ((int)blockIdx.x >= cnt) return;
        chain = rfl(a.tail_list[blockIdx.x]);
    } else if (chain >= a.n_chain) {
        return;
    }
    const int d = m.d;
    const bool lane_ok = lane < DP;

#ifndef LN_ROLE_ONLY   // tuning builds: one role's register needs on its own (-DLN_ROLE_ONLY=0 integrator, 1 bookkeeper, 2 matvec; the kernel hangs)
#define LN_ROLE_ONLY -1
#endif
    if (LN_ROLE_ONLY == 0 && w != 0) return;
    if (LN_ROLE_ONLY == 1 && w != LWV::KW) return;
    if (LN_ROLE_ONLY == 2 && (w < 1 || w >= LWV::KW)) return;
    if (w >= 1 && w < LWV::KW) {
        // ================================ job waves ================================
        LoneJobs<W, DEC, LWV::NT> jobs;
        jobs.load(m, LWV::ITILE ? w : w - 1, lane);
        for (;;) {
            __syncthreads();  // B0
            if (rfl(VI[LV_CMD]) == LN_DONE) break;
            __syncthreads();  // B1
            jobs.run(XT, GB);
            __syncthreads();  // B2
        }
        return;
    }

    if (w == 0) {
        // ================================ integrator ================================
        const double c_lin = lane_ok ? m.pd[PD_LIN * DP + lane] : 0.;
        const double c_mu = lane_ok ? m.pd[PD_MU * DP + lane] : 0.;
        const double c_dmu = (DEC == 1 && lane_ok) ? m.pd[PD_DMU * DP + lane] : 0.;
        const double c_smu = lane_ok ? m.pd[PD_SMU * DP + lane] : 0.;
        const int c_kind = (TR && lane_ok) ? (int)m.pd[PD_KIND * DP + lane] : 0;
        const double c_lo = (TR && lane_ok) ? m.pd[PD_LO * DP + lane] : 0., c_rg = (TR && lane_ok) ? m.pd[PD_RG * DP + lane] : 1.;
        double xs = 0., jac = 1., gj = 0., logdet_l = 0.;
        double q = 0., p = 0., g = 0., var = 1.;
        double eps = 0., eps_t = 0.;
        int dir = 1;
        double *tbl = TB + lane;
        const int xti = (lane & 3) * NS + (lane >> 2);   // x_dim -> row dim & 3, k-step dim >> 2 of the transposed operand
        LoneJobs<W, DEC, LWV::NT> jobs;   // (d <= 32: the integrator waits for the products anyway and takes a share of the jobs)
        if constexpr (LWV::ITILE) jobs.load(m, 0, lane);
        for (;;) {
            LTRACE(0, 0);
            __syncthreads();  // B0: the verdict on the leaf before the one just posted
            LTRACE(0, 1);
            // (the verdict in one round of LDS reads: every word is wanted before the first branch can be taken)
            const int4 vw = *(const int4 *)VI;
            const int2 vx = *(const int2 *)(VI + 4);
            const double v_eps_l = VD[0];
            const int cmd = rfl(vw.x), v_close = rfl(vw.y), v_eo = rfl(vw.z), v_eval = rfl(vw.w), v_dir = rfl(vx.x), v_init = rfl(vx.y);
            const double v_eps = rfl(v_eps_l);
            if (cmd == LN_DONE) break;
            bool evaluating = true;
            if (cmd == LN_NEW) {
                q = lane_ok ? NI[lane] : 0.;
                p = lane_ok ? NI[DP + lane] : 0.;
                g = lane_ok ? NI[2 * DP + lane] : 0.;
                var = lane_ok ? NI[3 * DP + lane] : 1.;
                eps = v_eps;
                dir = v_dir;
                eps_t = v_init ? 0. : eps * (double)dir;
            } else {
                int dir_use = dir;
                if (v_close) {
                    // the leaf in flight closes its doubling: park the new end (its p is the bookkeeper's to park) and start the
                    // next doubling in the direction the stream will give it (bf_nuts_pipe_kernel, phase A)
                    const int eo = v_eo;
                    if (lane_ok) { tbl[(eo + 0) * DP] = q; tbl[(eo + 2) * DP] = g; }
                    if (!v_eval) {
                        evaluating = false;
                    } else {
                        dir_use = v_dir;
                        if (dir_use != dir) {
                            const int eo2 = (dir_use > 0) ? SL_RIGHT_Q : SL_LEFT_Q;
                            q = lane_ok ? tbl[(eo2 + 0) * DP] : 0.;
                            p = lane_ok ? tbl[(eo2 + 1) * DP] : 0.;
                            g = lane_ok ? tbl[(eo2 + 2) * DP] : 0.;
                        }
                        dir = dir_use;
                    }
                }
                if (evaluating) eps_t = eps * (double)dir_use;
            }
            // ---- phase A: first half of the leapfrog step, B operands ----
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
                    logdet_l = 0. + LN_LOGV(fabs(J));
                    jac = J;
                    gj = J2 / J;
                }
                if (lane_ok) {
                    XT[xti] = xs;
                    XT[DP + xti] = xs - c_mu;
                    if constexpr (DEC == 1) XT[2 * DP + xti] = xs - c_dmu;
                }
            }
            LTRACE(0, 2);
            __syncthreads();  // B1
            LTRACE(0, 3);
            if constexpr (LWV::ITILE) jobs.run(XT, GB);
            LTRACE(0, 7);
            __syncthreads();  // B2
            LTRACE(0, 4);
            // ---- phase C: finish the evaluation (bf_nuts_pipe_kernel, phase C: the same sums in the same order) ----
            if (evaluating) {
                auto gb_read = [&](int slot_m) -> double {
                    const double *gp = GB + (slot_m * KS) * DP + lane;
                    double r = gp[0];
                    if (KS > 1) r += gp[DP];
                    return r;
                };
                const double sx = lane_ok ? gb_read(0) : 0.;
                const double hv = lane_ok ? gb_read(1) : 0.;
                const double dgr = DEC == 2 ? hv : ((DEC == 1 && lane_ok) ? gb_read(2) : 0.);   // H_decay (x - mu_decay); DEC = 2: the bound's product
                double gn = sx + c_lin;
                const double xm = xs - c_mu;
                constexpr bool fast_kin = !DEC;
                const double sv = sx - c_smu, gmu = c_smu + c_lin;
                double r_kin = 0., r_val, r_b2, r_bd2 = 0., r_a[2] = {0., 0.};
                if constexpr (DEC == 1) {
                    double r5[5] = {(xs - c_dmu) * dgr, __builtin_fma(0.5 * xs, sx, c_lin * xs), xm * hv, xm * gmu, xm * sv};
                    wave_sum_n<5>(r5);
                    r_bd2 = r5[0]; r_val = r5[1]; r_b2 = r5[2]; r_a[0] = r5[3]; r_a[1] = r5[4];
                } else if constexpr (DEC == 2) {   // (bf_nuts_pipe_kernel: the decay term's radius is the bound's)
                    double r4[4] = {__builtin_fma(0.5 * xs, sx, c_lin * xs), xm * hv, xm * gmu, xm * sv};
                    wave_sum_n<4>(r4);
                    r_val = r4[0]; r_b2 = r4[1]; r_a[0] = r4[2]; r_a[1] = r4[3];
                    r_bd2 = r_b2;
                } else {
                    double r3[3] = {0., __builtin_fma(0.5 * xs, sx, c_lin * xs), xm * hv};
                    {
                        double ge = gn * jac;
                        if constexpr (TR) ge += gj;
                        const double pe = p + (0.5 * eps_t) * ge;
                        r3[0] = pe * (var * pe);
                    }
                    wave_sum_n<3>(r3);
                    r_kin = r3[0]; r_val = r3[1]; r_b2 = r3[2];
                }
                LTRACE(0, 5);
                double logdet = 0.;
                if constexpr (TR) logdet = wave_sum(logdet_l);
                double f = (m.c0 + r_val) + 0.;
                double beta = 0.;
                const double a2 = m.alpha * m.alpha;
                if (!(r_b2 < a2 * (1. - 1e-12))) beta = LN_SQRT(r_b2);
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
                // (bf_nuts_pipe_kernel: finish)
                gn = gn * jac;
                if constexpr (DEC) {  // density.py:740-746
                    f -= m.decay_gamma * bf_clip0(r_bd2 - m.decay_alpha2);
                    if (r_bd2 > m.decay_alpha2) gn -= 2. * m.decay_gamma * dgr;
                }
                if constexpr (TR) {  // density.py:747-750
                    f += logdet;
                    gn += gj;
                }
                const double logp_new = f;
                const double dt = 0.5 * eps_t;
                p = p + dt * gn;        // integration.py:90
                g = gn;
                double kin = p * (var * p);   // metrics.py:88-91
                kin = kin_ready ? r_kin : wave_sum(kin);
                const double E_new = 0.5 * kin - logp_new;  // integration.py:92-93
                if (lane_ok) { LF[lane] = q; LF[DP + lane] = p; LF[2 * DP + lane] = g; }
                if (lane == 0) { LF[3 * DP] = E_new; LF[3 * DP + 1] = logp_new; }
            }
            LTRACE(0, 6);
#ifdef BF_LTRACE
            ltrip += 1;
#endif
        }
        return;
    }

    // ================================ bookkeeper ================================
    double *scp = a.sc + (size_t)chain * BFHIP_SC_N;
    double *vecp = a.vec + (size_t)chain * BFHIP_VEC_N * d;
    double *tbl = TB + lane;
    const int nw = a.cfg.n_warmup;
    auto load_vec = [&](int field, double pad) -> double { return (lane < d) ? vecp[field * d + lane] : pad; };
    auto store_vec = [&](int field, double v) { if (lane < d) vecp[field * d + lane] = v; };
    auto ldv = [&](int slot) -> double { return lane_ok ? tbl[slot * DP] : 0.; };
    auto stv = [&](int slot, double v) { if (lane_ok) tbl[slot * DP] = v; };
    auto lfv = [&](int k) -> double { return lane_ok ? LF[k * DP + lane] : 0.; };   // the posted leaf: q, p, g
    double q = 0., p = 0., g = 0., var = 1.;
    double TLp = 0., TPs = 0., TPq = 0., TRp = 0., L0p = 0., L0q = 0., TPg = 0., L0g = 0.;
    uint64_t rs[4], rs_save[4] = {0, 0, 0, 0};
    int i_iter, err;
    double eps = 0.;
    int dir = 1, depth = 0, i_leaf = 0, n_prop = 0, diverged = 0;
    double start_energy = 0., acc_sum = 0., T_W = 0., T_acc = 0., max_de = 0., w_off = 0., L0_W = 0., L0_acc = 0.;
    double E_pend = 0., lp_pend = 0.;
    double log_step, log_bar, hbar, smu, count, step_now, step_bar;
    double prop_E = 0., prop_logp = 0., tree_W = 0., T_E = 0., T_logp = 0.;
    unsigned long long nlf = 0;
    int prev = LF_NONE, fly = LF_NONE;
    bool kdone = false;
    int kb = 0;
    // step_size.py:31-45: what the update at the iteration's end needs of `count` alone -- sqrt(count), count ** -k, 1 / (count + t_0)
    // -- is taken in the window after a tree starts, when this wave has nothing to account for (the same expressions, earlier)
    double ad_sq = 0., ad_mk = 0., ad_wgt = 0.;
    bool ad_ready = false;
    double inv_sd = 0.;   // 1 / sqrt(var): metrics.py:83-86 divides every draw by it; it changes when the metric does
    double *WF = lds + LG::o_WF;   // the Welford windows of the metric (metrics.py:333-371) staged in LDS, written through to the state
#define KBAR() do { if (kb < 2) { __syncthreads(); ++kb; } } while (0)

    // metric.random: samplers/hmc_utils/metrics.py:83-86 (the stream layout of every sampler kernel)
    auto draw_momentum = [&]() {
        const uint64_t K = bf_xoshiro_next(rs);
        const uint64_t P = (uint64_t)(lane >> 1);
        const double u1 = bf_u01_open0(bf_mix64(K + (2 * P + 1) * BF_GOLDEN));
        const double u2 = bf_u01(bf_mix64(K + (2 * P + 2) * BF_GOLDEN));
        const double rad = LN_SQRTV(-2. * LN_LOGV(u1));
        double sn, cs;
        sincospi(ln_opq(2. * u2), &sn, &cs);
        sn = ln_opq(sn); cs = ln_opq(cs);
        const double z = (lane & 1) ? rad * sn : rad * cs;
        p = (lane < d) ? inv_sd * z : 0.;
        g = 0.;
    };
    auto adapt_consts = [&]() {
        if (!ad_ready && i_iter < nw && a.cfg.adapt_step_size) {
            ad_wgt = 1. / (count + a.cfg.t_0);
            ad_sq = LN_SQRT(count);
            ad_mk = LN_EXP(-a.cfg.k * LN_LOG(count));  // count ** -k
            ad_ready = true;
        }
    };
    // a new tree from (q, p, g): the verdict the integrator starts it with
    auto post_new = [&](int init) {
        if (lane_ok) { NI[lane] = q; NI[DP + lane] = p; NI[2 * DP + lane] = g; NI[3 * DP + lane] = var; }
        if (lane == 0) {
            VD[0] = eps;
            VI[LV_CMD] = LN_NEW; VI[LV_CLOSE] = 0; VI[LV_EO] = 0; VI[LV_EVAL] = 1; VI[LV_DIR] = dir; VI[LV_INIT] = init;
        }
    };
    // Tree.__init__ (nuts.py:24-43) on the state (q, p, g); the direction of the first doubling has been drawn (rs_save: the
    // stream before that draw, restored when the energy is bad -- the reference raises before it would draw)
    auto init_tree = [&](double E0, double logp0) -> bool {
        if (!(fabs(E0) <= 1.7976931348623157e308)) {
            err = 1;
            for (int k = 0; k < 4; ++k) rs[k] = rs_save[k];
            return false;
        }
        start_energy = E0;
        stv(SL_LEFT_Q, q); stv(SL_LEFT_P, p); stv(SL_LEFT_G, g);
        stv(SL_RIGHT_Q, q); stv(SL_RIGHT_P, p); stv(SL_RIGHT_G, g);
        stv(SL_PROP_Q, q); stv(SL_PSUM, p);
        stv(SL_PROPG, g);
        prop_E = E0;
        prop_logp = logp0;
        tree_W = 1.;
        w_off = 0.;
        max_de = 0.;
        depth = 0; acc_sum = 0.; n_prop = 0; diverged = 0; i_leaf = 0;
        return true;
    };
    auto draw_dir = [&]() {
        for (int k = 0; k < 4; ++k) rs_save[k] = rs[k];
        dir = (bf_u01(bf_xoshiro_next(rs)) < 0.5) ? 1 : -1;  // nuts.py:210, log(U) < log(1/2)
    };

    for (int k = 0; k < 4; ++k) rs[k] = rfl((uint64_t)a.rng[(size_t)chain * 4 + k]);
    log_step = rfl(scp[BFHIP_SC_LOG_STEP]);
    log_bar = rfl(scp[BFHIP_SC_LOG_BAR]);
    hbar = rfl(scp[BFHIP_SC_HBAR]);
    smu = rfl(scp[BFHIP_SC_MU]);
    count = rfl(scp[BFHIP_SC_COUNT]);
    step_now = LN_EXP(log_step);
    step_bar = LN_EXP(log_bar);
    i_iter = rfl((int)scp[BFHIP_SC_I_ITER]);
    err = rfl((int)scp[BFHIP_SC_ERROR]);
    q = load_vec(BFHIP_VEC_Q, 0.);
    var = load_vec(BFHIP_VEC_VAR, 1.);
    inv_sd = 1. / LN_SQRTV(var);
    const bool wf_on = i_iter < nw && a.cfg.adapt_metric;
    if (wf_on) {
        if (lane_ok) {
            WF[lane] = load_vec(BFHIP_VEC_FG_MEAN, 0.); WF[DP + lane] = load_vec(BFHIP_VEC_FG_RAW, 0.);
            WF[2 * DP + lane] = load_vec(BFHIP_VEC_BG_MEAN, 0.); WF[3 * DP + lane] = load_vec(BFHIP_VEC_BG_RAW, 0.);
        }
        if (lane < 5) WF[4 * DP + lane] = scp[BFHIP_SC_FG_N + lane];   // fg_n, bg_n, n_samples, previous_update, adapt_window
    }
    if (i_iter < a.iter_end && err == 0) {
        draw_momentum();
        eps = (i_iter < nw) ? step_now : step_bar;
        draw_dir();
        post_new(1);   // compute_state at the start of a launch (base_hmc.py:70): a step of length 0
        fly = LF_INIT;
    } else {
        if (lane == 0) VI[LV_CMD] = LN_DONE;
        kdone = true;
    }

    for (;;) {
        LTRACE(1, 0);
        __syncthreads();  // B0
        LTRACE(1, 1);
        if (kdone) break;
        kb = 0;
        bool ended = false;
        if (prev == LF_INIT) {
            // the evaluation that opened the launch: BaseHMC.astep start, base_hmc.py:70-76
            q = lfv(0); p = lfv(1); g = lfv(2);
            const double E0 = rfl(LF[3 * DP]), logp0 = rfl(LF[3 * DP + 1]);
            KBAR(); KBAR();   // (the integrator parks nothing this trip; the tree's slots are written behind its phase A anyway)
            if (!init_tree(E0, logp0)) { kdone = true; }
        } else if (prev == LF_LEAF) {
            {   // (one round of LDS reads)
                const double lq_ = lfv(0), lp_ = lfv(1), lg_ = lfv(2);
                const double le_ = LF[3 * DP], ll_ = LF[3 * DP + 1];
                TPq = lq_; TRp = lp_; TPg = lg_;
                E_pend = rfl(le_);
                lp_pend = rfl(ll_);
            }
            LTRACE(1, 4);
            int unit = U_EVAL, lev = 0;
            // ---- Tree._single_step: nuts.py:105-132 ----
            nlf += 1;
            n_prop += 1;
            double dE = E_pend - start_energy;
            if (dE != dE) dE = INFINITY;
            if (fabs(dE) > fabs(max_de)) max_de = dE;
            T_E = E_pend;
            T_logp = lp_pend;
            T_acc = 0.;
            if (!(fabs(dE) < a.cfg.max_change)) {
                diverged = 1;
                unit = U_ABORT;
            }
            LTRACE(1, 11);
            KBAR();
            LTRACE(1, 10);
            // One block for the two long chains of a leaf, so that the scheduler runs them side by side: the exponential of the
            // multinomial weight (a dependent chain of ~25 FP64 instructions, 32-40 cycles each) and the U-turn sums of the level-0
            // merge (nuts.py:150-151: they need the momenta only).  Both are taken whether or not they will be used.
            double aw = -dE - w_off;
            const double e_aw = LN_EXPV(aw);
            const double ps0 = L0p + TRp;
            double r2[2] = {ps0 * (var * L0p), ps0 * (var * TRp)};
            wave_sum_n<2>(r2);
            if (unit == U_EVAL) {
                // multinomial weight exp(-dE) relative to a running offset w_off (exact streaming log-sum-exp)
                T_W = rfl(e_aw);
                if (aw > 600.) {
                    const double sc_ = LN_EXP(-aw);
                    tree_W = tree_W * sc_;
                    if (lane == 0)
                        for (int l2 = 0; l2 < depth; ++l2) LS[l2 * LS_N + LS_LS] *= sc_;
                    L0_W *= sc_;
                    w_off = w_off + aw;
                    aw = 0.;
                    T_W = 1.;   // exp(0)
                }
                const double pacc = (w_off == 0.) ? T_W : LN_EXP(-dE);
                T_acc = pacc > 1. ? 1. : pacc;
                TLp = TRp;
                TPs = TRp;
                unit = U_MERGE;
            }
            LTRACE(1, 5);
            if (unit == U_MERGE && (i_leaf & 1) && depth > 0) {
                // ---- level-0 merge with the previous leaf (nuts.py:146-178) ----
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
                        T_E = rfl(LS[LS_E]);
                        T_logp = rfl(LS[LS_LOGP]);
                    }
                    T_W = Wsum;
                    TPs = L0p + TRp;
                    TLp = L0p;
                }
            }
            LTRACE(1, 6);
            KBAR();
            LTRACE(1, 7);
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
                const double *lsp = LS + lev * LS_N;
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
                        T_E = rfl(lsp[LS_E]);
                        T_logp = rfl(lsp[LS_LOGP]);
                    }
                    T_W = Wsum;
                    TLp = A;
                    TPs = psum;
                }
            }
            if (unit == U_MERGE) {
                if (lev < depth) {
                    // the subtree waits for its right sibling
                    if (lev == 0) {
                        L0p = TRp;
                        L0q = TPq;
                        L0g = TPg;
                        L0_W = T_W;
                        L0_acc = T_acc;
                    } else {
                        const int slot = SL_STACK + 4 * lev;
                        stv(slot + 0, TLp); stv(slot + 1, TRp); stv(slot + 2, TPs); stv(slot + 3, TPq);
                        stv(SL_PG + lev, TPg);
                    }
                    if (lane == 0) {
                        double *lsp = LS + lev * LS_N;
                        lsp[LS_LS] = T_W; lsp[LS_ACC] = T_acc;
                        lsp[LS_E] = (lev == 0) ? E_pend : T_E;
                        lsp[LS_LOGP] = (lev == 0) ? lp_pend : T_logp;
                    }
                    i_leaf += 1;
                    unit = U_DONE;
                } else {
                    unit = U_DBL_END;
                }
            }
            LTRACE(1, 8);
            if (unit == U_ABORT) {
                // unwind: every pending ancestor adds its left half's accept_sum (nuts.py:173)
                for (int al = (diverged ? 0 : lev); al < depth; ++al)
                    if ((i_leaf >> al) & 1) T_acc = rfl(LS[al * LS_N + LS_ACC]) + T_acc;
                depth += 1;  // nuts.py:71-73
                acc_sum += T_acc;
                unit = U_END1;
            } else if (unit == U_DBL_END) {
                // ---- Tree.extend after a complete subtree: nuts.py:71-103 ----
                // (the integrator has parked the q and the gradient of the new end; its p is TRp)
                double ps = ldv(SL_PSUM);
                const double oldL = ldv(SL_LEFT_P), oldR = ldv(SL_RIGHT_P);
                depth += 1;
                acc_sum += T_acc;
                {   // :81-83  logbern(ls_new - ls_old)  <=>  U * W_old < W_new
                    if (T_W != T_W || tree_W != tree_W) err = 2;
                    const double u = bf_u01(bf_xoshiro_next(rs));
                    if ((u * tree_W < T_W) || (u == 0.)) {
                        stv(SL_PROP_Q, TPq);
                        stv(SL_PROPG, TPg);
                        prop_E = T_E;
                        prop_logp = T_logp;
                    }
                    tree_W = tree_W + T_W;  // :85
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
                    dir = (bf_u01(bf_xoshiro_next(rs)) < 0.5) ? 1 : -1;  // nuts.py:210 (announced one leaf ago)
                    i_leaf = 0;
                    unit = U_DONE;
                }
            }
            LTRACE(1, 9);
            if (__builtin_expect(unit == U_END1, 0)) {   // (cold: once per tree; the register allocator keeps its spill code here)
                ended = true;  // the tree the leaf in flight belongs to has ended: that leaf is dropped
                if (err == 0) {
                    // the proposal becomes the sample and the start of the next iteration (value and gradient travelled with it)
                    const double g_prop = ldv(SL_PROPG);
                    q = ldv(SL_PROP_Q);
                    // ================= iteration end (base_hmc.py:80-85) =================
                    const bool warm = i_iter < nw;
                    const double accept_stat = acc_sum / (double)n_prop;  // nuts.py:186
                    if (warm && a.cfg.adapt_step_size) {  // step_size.py:31-45
                        adapt_consts();   // (normally taken long ago)
                        const double wgt = ad_wgt;
                        // (the contractions below are bf_nuts_pipe_kernel's, spelled out: left to the compiler this kernel fused the other
                        // product of each sum and its step sizes came out an ulp off)
                        hbar = __builtin_fma(wgt, (a.cfg.target_accept - accept_stat), (1. - wgt) * hbar);
                        log_step = smu - hbar * ad_sq / a.cfg.gamma;
                        const double mk = ad_mk;
                        log_bar = __builtin_fma(mk, log_step, (1. - mk) * log_bar);
                        count = count + 1.;
                        ad_ready = false;
                        const double e1 = LN_EXPV(log_step), e2 = LN_EXPV(log_bar);   // (one block: side by side)
                        step_now = rfl(e1);
                        step_bar = rfl(e2);
                    }
                    const int orow = i_iter - a.iter_out0;
                    if (orow >= 0 && orow < a.n_out) {
                        if (lane == 0) {
                            double *st = a.stats + ((size_t)chain * a.n_out + orow) * BFHIP_STAT_STRIDE;
                            st[BFHIP_NS_LOGP] = prop_logp;
                            st[BFHIP_NS_ENERGY] = prop_E;
                            st[BFHIP_NS_TREE_DEPTH] = (double)depth;
                            st[BFHIP_NS_TREE_SIZE] = (double)n_prop;
                            st[BFHIP_NS_MEAN_TREE_ACCEPT] = accept_stat;
                            st[BFHIP_NS_STEP_SIZE] = step_now;
                            st[BFHIP_NS_STEP_SIZE_BAR] = step_bar;
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
                        double fg_n = rfl(WF[4 * DP + 0]), bg_n = rfl(WF[4 * DP + 1]);
                        double n_samples = rfl(WF[4 * DP + 2]), prev_upd = rfl(WF[4 * DP + 3]);
                        double adapt_window = rfl(WF[4 * DP + 4]);
                        const long delta = (long)(n_samples - prev_upd);
                        double fm = lane_ok ? WF[lane] : 0., fr = lane_ok ? WF[DP + lane] : 0.;
                        double bm = lane_ok ? WF[2 * DP + lane] : 0., br = lane_ok ? WF[3 * DP + lane] : 0.;
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
                            inv_sd = 1. / LN_SQRTV(var);
                        }
                        if ((double)delta >= adapt_window) {
                            fm = bm; fr = br; bm = 0.; br = 0.;
                            fg_n = bg_n;
                            bg_n = 10.;
                            prev_upd = n_samples;
                            if (a.cfg.doubling) adapt_window *= 2.;
                        }
                        n_samples += 1.;
                        if (lane_ok) { WF[lane] = fm; WF[DP + lane] = fr; WF[2 * DP + lane] = bm; WF[3 * DP + lane] = br; }
                        store_vec(BFHIP_VEC_FG_MEAN, fm);
                        store_vec(BFHIP_VEC_FG_RAW, fr);
                        store_vec(BFHIP_VEC_BG_MEAN, bm);
                        store_vec(BFHIP_VEC_BG_RAW, br);
                        if (lane == 0) {
                            WF[4 * DP + 0] = fg_n; WF[4 * DP + 1] = bg_n; WF[4 * DP + 2] = n_samples; WF[4 * DP + 3] = prev_upd; WF[4 * DP + 4] = adapt_window;
                            scp[BFHIP_SC_FG_N] = fg_n;
                            scp[BFHIP_SC_BG_N] = bg_n;
                            scp[BFHIP_SC_N_SAMPLES] = n_samples;
                            scp[BFHIP_SC_PREV_UPDATE] = prev_upd;
                            scp[BFHIP_SC_ADAPT_WINDOW] = adapt_window;
                        }
                    }
                    i_iter += 1;
                    if (i_iter < a.iter_end) {
                        draw_momentum();
                        g = g_prop;
                        const double logp0 = prop_logp;
                        const double kin0 = wave_sum(p * (var * p));   // metrics.py:88-91
                        eps = (i_iter < nw) ? step_now : step_bar;     // step_size.py:25-29
                        KBAR(); KBAR();   // (the tree's slots and the start buffers are written behind the integrator's phase A)
                        if (init_tree(0.5 * kin0 - logp0, logp0)) {  // integration.py:28-34
                            draw_dir();
                            post_new(0);
                        } else {
                            kdone = true;
                        }
                    } else {
                        kdone = true;
                    }
                }
            }
            if (err != 0) { kdone = true; ended = true; }
        }
        if (prev == LF_NONE) adapt_consts();   // (nothing to account for: the leaf in flight is a tree's first)
        KBAR(); KBAR();
        LTRACE(1, 2);
        if (kdone) {
            if (lane == 0) VI[LV_CMD] = LN_DONE;
        } else if (ended) {
            prev = LF_NONE;
            fly = LF_LEAF;
        } else {
            // the verdict on the leaf in flight: go on; where its doubling ends and which way the next one goes
            prev = fly;
            int close = 0, eo = 0, ev = 1, dir_use = dir;
            if (fly == LF_LEAF && i_leaf == (1 << depth) - 1) {
                close = 1;
                eo = (dir > 0) ? SL_RIGHT_Q : SL_LEFT_Q;
                if (depth + 1 >= a.cfg.max_treedepth) {
                    ev = 0;  // the tree stops at this depth whatever the checks say
                } else {
                    uint64_t t[4] = {rs[0], rs[1], rs[2], rs[3]};
                    for (int k = 0; k <= depth; ++k) (void)bf_xoshiro_next(t);  // `depth` merges and the swap
                    dir_use = (bf_u01(bf_xoshiro_next(t)) < 0.5) ? 1 : -1;
                }
            }
            if (lane == 0) {
                VI[LV_CMD] = LN_CONT; VI[LV_CLOSE] = close; VI[LV_EO] = eo; VI[LV_EVAL] = ev; VI[LV_DIR] = dir_use;
            }
            fly = ev ? LF_LEAF : LF_NONE;
        }
        LTRACE(1, 3);
#ifdef BF_LTRACE
        ltrip += 1;
#endif
    }
#undef KBAR

#ifdef BF_LTRACE
    if (a.stamps && blockIdx.x == 0)
        for (int i = lane; i < BF_LTRACE * 32; i += 64) a.stamps[i] = LTRC[i];
#endif
    // ---- write the chain state back ----
    if (err == 2 && lane_ok) q = LF[lane];   // (the pipelined kernel leaves the position of the leaf in flight)
    store_vec(BFHIP_VEC_Q, q);
    if (lane == 0) {
        for (int k = 0; k < 4; ++k) a.rng[(size_t)chain * 4 + k] = rs[k];
        scp[BFHIP_SC_LOG_STEP] = log_step;
        scp[BFHIP_SC_LOG_BAR] = log_bar;
        scp[BFHIP_SC_HBAR] = hbar;
        scp[BFHIP_SC_COUNT] = count;
        scp[BFHIP_SC_I_ITER] = (double)i_iter;
        scp[BFHIP_SC_ERROR] = (double)err;
        if (a.n_leapfrog && nlf) atomicAdd(a.n_leapfrog, nlf);
    }
}
#undef LN_EXP
#undef LN_LOG
#undef LN_SQRT
#undef LN_EXPV
#undef LN_LOGV
#undef LN_SQRTV
#pragma pop_macro("exp")
#pragma pop_macro("log")
#pragma pop_macro("sqrt")
#pragma pop_macro("sincospi")
