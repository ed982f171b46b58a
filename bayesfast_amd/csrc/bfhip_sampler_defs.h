// bfhip_sampler_defs.h -- launch arguments and state-machine constants shared by the sampler kernels.
#pragma once
#include "bfhip_model.h"

struct SamplerArgs {
    bfhip_sampler_config cfg;
    int n_chain, iter_end, iter_out0, n_out, nslot;
    int cpg;       // wave-per-chain kernels: chains per workgroup (0: one per wave), bfhip_sampler.hip: wave_layout_cpg
    int no_quad;   // bf_sampler_kernel: 16-column tiles even with at most four chains in the workgroup (bfhip_debug_set("no_quad"): tests)
    int cub_lds;   // bf_sampler_kernel: the cubic coefficient tables are staged in LDS (sampler_cubic_lds)
    int pld_cl;    // bf_sampler_kernel, pipeline density, eight-wave forms: a row-major copy of C' is staged in LDS (PldLds::CL)
    int tail_max;  // plain kernel: at most this many evaluating chains of a group take the VALU matvec (0: never)
    int ks, gbn;  // K-split of the matvec jobs (so that all 16 waves get one) and the number of result slots
    uint64_t *rng;
    double *sc, *vec, *samples, *stats;
    unsigned long long *n_leapfrog;
    double *scratch;
    double *mat;  // full-rank metric: [n_chain][BF_MAT_N][d][d] (transposed storage, bfhip_metric.h), or NULL
    unsigned long long *gcount;  // group kernel, measurement only: [0] += trips, [1] += trips that ran the bound's tiles, [2] += trips with a late exchange, [3] += trips without the early one (4 words, or NULL)
    int no_bound_proof;          // group kernel, tests only: always compute the H (x - mu) tiles (bfhip_debug_set("no_bound_proof"))
    // bf_nuts_pipe_kernel, the launch's tail (bfhip_sampler.hip: launch_nuts_pipe): tail_stop > 0 -- a workgroup with at most that
    // many unfinished chains lets each of them stop at the end of its iteration; tail_list / tail_count -- the chains of THIS launch
    // (the ones that stopped), tail_count[0] of them, one to four per workgroup (n_cu decides)
    int tail_stop, n_cu;
    int tail_q;
    int cub_loops;      // cubic configs: the general loops also at sixteen masked inputs (tests compare; BFHIP_CUBIC_LOOPS)       // ... once tail_q quarters of the launch's chains are through
    const int *tail_list, *tail_count;
    int *tail_done;   // first part: += 1 per chain that has finished the launch's iterations (a workgroup stops its last chains only
                      // when three quarters of the launch's chains are through: the tail, not a slow workgroup among busy ones)
    unsigned long long *stamps;  // diagnostics only: [groups][16 waves][20]: 10 cycle counters + 10 event counts, or NULL
};

enum { M_INIT = 0, M_LEAF = 1, M_OOB = 2, M_DONE = 3 };
// units of per-chain work; a chain runs one per trip (U_EVAL needs this trip's gradient)
enum { U_EVAL = 0, U_MERGE_RUN, U_DBL_END, U_END1, U_END2, U_END3, U_DONE, U_MERGE, U_ABORT };
enum { SL_LEFT_Q = 0, SL_LEFT_P, SL_LEFT_G, SL_RIGHT_Q, SL_RIGHT_P, SL_RIGHT_G, SL_PROP_Q, SL_PSUM, SL_STACK };
enum { LS_LS = 0, LS_E, LS_LOGP, LS_ACC, LS_N };
// cold per-chain scalars parked in LDS (one writer: lane 0 of the chain's wave; broadcast reads)
enum { CS_LOG_STEP = 0, CS_LOG_BAR, CS_HBAR, CS_SMU, CS_COUNT, CS_PROP_E, CS_PROP_LOGP, CS_MAX_DE, CS_HACC, CS_HDE,
       CS_W_OFF, CS_TREE_W, CS_BETA, CS_T_E, CS_T_LOGP, CS_STEP_NOW, CS_STEP_BAR, CS_N };


// bfhip_sampler.hip: LDS bytes of bf_sampler_kernel's own regions for a pipeline density (bfhip_pld.hip sizes its block with it)
size_t bf_sampler_lds_bytes_base(const DevModel &m);
// bfhip_pld.hip: Density.logp_and_grad of the pipeline density
int bf_pld_logp_grad(struct bfhip_ctx *ctx, int n, const double *x, int original_space, double *logp, double *grad);
// bfhip_group.hip: NUTS / HMC for the common surrogate (optionally with decay / constraint transform) at d <= 64
struct bfhip_ctx;
bool bf_group_supports(const DevModel &m, const SamplerArgs &args);
int bf_launch_group(bfhip_ctx *ctx, const SamplerArgs &args);
int bf_no_bound_proof();  // test hook state (bfhip_debug_set("no_bound_proof"))
// bfhip_split.h: NUTS on the plain common surrogate at 33 <= d <= 64, integrator and bookkeeper waves (chain_layout 3)
bool bf_split_supports(const DevModel &m, const SamplerArgs &args);
int bf_launch_split(bfhip_ctx *ctx, const SamplerArgs &args);
