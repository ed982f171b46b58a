// bfhip_common.h -- shared device/host definitions of libbfhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bfhip.h"
#include "bfhip_model.h"
#include "bfhip_tune.h"

// ---- wave helpers ---------------------------------------------------------------------------------
__device__ inline double bf_shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }

// numpy.logaddexp
__device__ inline double bf_logaddexp(double x, double y) {
    if (x == y) return x + 0.6931471805599453094;
    double tmp = x - y;
    if (tmp > 0) return x + log1p(exp(-tmp));
    if (tmp <= 0) return y + log1p(exp(tmp));
    return tmp;  // NaN
}

#define BF_HIP_CHECK(expr)                                                                 \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) return bf_set_error(BFHIP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

int bf_set_error(int code, const char *fmt, ...);

struct bfhip_ctx {
    int device;
    hipStream_t stream;
    int has_model;
    DevModel model;
    void *model_buf;      // one device allocation holding pd + fragments
    size_t model_bytes;
    void *cubic_buf;      // cubic-term tables
    size_t cubic_bytes;
    void *pm_buf;         // multi-output polymodel (bfhip_polymodel_upload)
    size_t pm_bytes;
    int has_pm;
    void *pld_buf;        // pipeline density (bfhip_pipeline_upload): fragments, tables
    size_t pld_bytes;
    struct PolyDev {
        int d, DP, m, use_bound, has_quad;
        const double *Sf;    // [m][DP*DP] A fragments of S_o = A_o + A_o^T
        const double *lin;   // [m][DP]
        const double *c0;    // [m]
        const double *f_mu;  // [m]
        const double *mu;    // [DP]
        const double *Hf;    // [DP*DP]
        double alpha;
        // cubic configs, compact over their masks (per output o: A2[o][n2][n2], A2t[o], T3t[o][n3][n3][n3] as in DevModel)
        int n2, n3;
        const int *mask2, *pos2, *mask3, *pos3;
        const double *A2, *A2t, *T3t;
    } pm;
    int *tail_buf;        // the chains of a launch's tail: count, then their indices (bfhip_sampler.hip: launch_nuts_pipe)
    int tail_cap;
    void *scratch;        // sampler tree scratch (grow-only)
    size_t scratch_bytes;
    int n_cu;
    void *flow;           // counters and exchange buffers of the triangular solves (bfhip_fit.hip: ensure_flow)
    int flow_cap;         // in 64-row blocks
};

// Every entry point that launches or allocates runs on its context's device, whatever the caller's current device is
// (one process may hold contexts of several GPUs); the previous device is restored on return.
struct BfDeviceGuard {
    int prev = -1;
    explicit BfDeviceGuard(const bfhip_ctx *ctx) {
        if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device) (void)hipSetDevice(ctx->device);
        else prev = -1;
    }
    ~BfDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// the common surrogate: linear + quadratic configs with the extrapolation bound and nothing else
static inline bool bf_model_plain(const DevModel &m) {
    return m.has_quad && m.use_bound && !m.use_decay && !m.has_transform && !m.has_su && !m.has_cubic && !m.has_link;
}
