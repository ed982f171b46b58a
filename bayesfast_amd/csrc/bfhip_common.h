// bfhip_common.h -- shared device/host definitions of libbfhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bfhip.h"

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// ---- per-dimension parameter table (DP doubles each) ---------------------------------------------
enum {
    PD_KIND = 0,  // constraint kind as a double: 0 affine, 1 both bounds, 2 lower only, 3 upper only
    PD_LO,        // ranges[:,0]
    PD_RG,        // ranges[:,1] - ranges[:,0]
    PD_SU_LO,     // Surrogate.input_scales[:,0]
    PD_SU_DIFF,   // Surrogate._input_scales_diff
    PD_LIN,       // linear coefficients
    PD_MU,        // bound centre
    PD_DMU,       // decay centre
    PD_N
};

// Device-resident surrogate density.  Matrices are stored as MFMA A-operand fragments:
//   frag[(t * NS + s) * 64 + l] = M[16 t + (l & 15)][4 s + (l >> 4)],  t < T = DP/16, s < NS = DP/4
// so that one wave-instruction reads 512 contiguous bytes (v_mfma_f64_16x16x4_f64, A[i=l&15][k=l>>4]).
struct DevModel {
    int d, DP;
    int has_transform, has_su, has_quad, use_bound, use_decay, has_cubic;
    const double *pd;   // [PD_N][DP]
    const double *Sf;   // quadratic form, symmetrised: S = A + A^T (so grad = S x + lin, f = c0 + lin.x + x.Sx/2)
    const double *Hf;   // bound Hessian
    const double *Hdf;  // decay Hessian
    double c0, alpha, f_mu, decay_alpha2, decay_gamma;
    // cubic terms in compact (masked) form; pos2/pos3 map a dimension to its index in the mask or -1
    int n2, n3;
    const int *mask2, *pos2, *mask3, *pos3;  // mask: [n], pos: [DP]
    const double *A2;    // [n2][n2] cubic-2 coefficients a[j][k]  (f = sum_j x_j^2 sum_k a[j][k] x_k)
    const double *A2t;   // [n2][n2] its transpose
    const double *T3t;   // [n3][n3][n3] symmetric fill of the j<k<l coefficients, zero where indices repeat,
                         // stored [k][l][j] so that consecutive lanes (j) read consecutive words
};

// ---- xoshiro256++ / splitmix64 --------------------------------------------------------------------
__host__ __device__ inline uint64_t bf_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

__host__ __device__ inline uint64_t bf_xoshiro_next(uint64_t (&s)[4]) {
    uint64_t result = bf_rotl(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = bf_rotl(s[3], 45);
    return result;
}

__host__ __device__ inline uint64_t bf_mix64(uint64_t z) {  // splitmix64 output function
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define BF_GOLDEN 0x9E3779B97F4A7C15ULL
#define BF_TWO_M53 1.1102230246251565e-16
#define BF_TWO_PI 6.283185307179586476925286766559

__host__ __device__ inline double bf_u01(uint64_t x) { return (double)(x >> 11) * BF_TWO_M53; }          // [0,1)
__host__ __device__ inline double bf_u01_open0(uint64_t x) { return (double)((x >> 11) + 1) * BF_TWO_M53; } // (0,1]

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
    struct PolyDev {
        int d, DP, m, use_bound, has_quad;
        const double *Sf;    // [m][DP*DP] A fragments of S_o = A_o + A_o^T
        const double *lin;   // [m][DP]
        const double *c0;    // [m]
        const double *f_mu;  // [m]
        const double *mu;    // [DP]
        const double *Hf;    // [DP*DP]
        double alpha;
    } pm;
    void *scratch;        // sampler tree scratch (grow-only)
    size_t scratch_bytes;
    int n_cu;
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
    return m.has_quad && m.use_bound && !m.use_decay && !m.has_transform && !m.has_su && !m.has_cubic;
}
