// bfhip_fit.hip -- surrogate fit on device: design-matrix blocks, normal equations on FP64 MFMA, SPD solve.
//
// PolyModel.fit (modules/poly.py:505-589) builds A = [1 | x | quadratic | cubic-2 | cubic-3] per output and
// calls scipy.linalg.lstsq (LAPACK gelsd) on it, once per output.  Here the design blocks are written by a
// memory-bound kernel, G = A^T A is a genuine dense contraction and runs on v_mfma_f64_16x16x4_f64, and the
// P x P system is solved by a blocked Cholesky with Jacobi equilibration (all outputs share one
// factorisation).
#include <vector>
#include "bfhip_common.h"

__device__ inline double bf_readlane(double v, int l) {  // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// ---------------------------------------------------------------------------------------------------
// design blocks: modules/_poly.pyx:143-177 (+ the [1 | x] block of modules/poly.py:537-543)
// ---------------------------------------------------------------------------------------------------
__global__ void bf_design_block_kernel(int order, int n, int n_in, const double *__restrict__ x,
                                       const double *__restrict__ w, double *__restrict__ A, int lda, int col0,
                                       int width) {
    extern __shared__ double xr[];  // the row's inputs
    const int i = blockIdx.x;
    if (i >= n) return;
    for (int k = threadIdx.x; k < n_in; k += blockDim.x) xr[k] = x[(size_t)i * n_in + k];
    __syncthreads();
    const double wi = w ? w[i] : 1.;
    double *out = A + (size_t)i * lda + col0;
    for (int j = threadIdx.x; j < width; j += blockDim.x) {
        double v;
        if (order == 0) {
            v = j == 0 ? 1. : xr[j - 1];
        } else if (order == 1) {
            // j-th pair (k <= l) in row-major order of the upper triangle: offset(k) = k n - k (k - 1) / 2
            int k = (int)(((2. * n_in + 1.) - sqrt((2. * n_in + 1.) * (2. * n_in + 1.) - 8. * j)) * 0.5);
            if (k < 0) k = 0;
            while (k > 0 && k * n_in - k * (k - 1) / 2 > j) --k;
            while ((k + 1) * n_in - (k + 1) * k / 2 <= j) ++k;
            const int l = k + (j - (k * n_in - k * (k - 1) / 2));
            v = xr[k] * xr[l];
        } else if (order == 2) {
            const int k = j / n_in, l = j % n_in;
            v = xr[k] * xr[k] * xr[l];
        } else {
            // j-th triple k < l < p in lexicographic order
            int rem = j, k = 0;
            for (;; ++k) {
                const int cnt = (n_in - k - 1) * (n_in - k - 2) / 2;
                if (rem < cnt) break;
                rem -= cnt;
            }
            int l = k + 1;
            for (;; ++l) {
                const int cnt = n_in - l - 1;
                if (rem < cnt) break;
                rem -= cnt;
            }
            const int p = l + 1 + rem;
            v = xr[k] * xr[l] * xr[p];
        }
        out[j] = v * wi;  // optional row weights, modules/poly.py:566-568
    }
}

extern "C" int bfhip_design_block(bfhip_ctx *ctx, int order, int n, int n_in, const double *x, const double *w,
                                  double *A, int lda, int col0) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || order < 0 || order > 3 || n < 0 || n_in < 1 || n_in > 1024 || lda < 1 || col0 < 0)
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_design_block: invalid argument");
    if (n == 0) return 0;
    if (!x || !A) return bf_set_error(BFHIP_ERR_ARG, "bfhip_design_block: NULL array");
    long width = order == 0 ? n_in + 1 : order == 1 ? (long)n_in * (n_in + 1) / 2 : order == 2 ? (long)n_in * n_in
                                                                                             : (long)n_in * (n_in - 1) * (n_in - 2) / 6;
    if (col0 + width > lda) return bf_set_error(BFHIP_ERR_ARG, "bfhip_design_block: block does not fit in lda");
    if (width == 0) return 0;
    hipLaunchKernelGGL(bf_design_block_kernel, dim3(n), dim3(256), n_in * sizeof(double), ctx->stream, order, n, n_in, x,
                       w, A, lda, col0, (int)width);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Gram matrix on MFMA.  One wave owns a 64 x 64 block (bi <= bj) of G and a slice of the rows (split-K);
// the A operand of v_mfma_f64_16x16x4_f64 is A^T read in place (lane (i = l&15, k = l>>4) loads
// A[r0 + k][I + i]: 16 consecutive columns of 4 consecutive rows), the B operand the same pattern at J.
// Partials are summed in a fixed order by a second kernel (bitwise reproducible, no atomics).
// ---------------------------------------------------------------------------------------------------
#define GB_ 64
__global__ __launch_bounds__(256) void bf_gram_kernel(int n, int P, const double *__restrict__ A, int lda, int nb,
                                                      int split, double *__restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_blk = nb * (nb + 1) / 2;
    if (wid >= n_blk * split) return;
    const int blk = wid / split, sk = wid % split;
    // decode (bi <= bj) from the linear upper-triangle index
    int bi = 0, rem = blk;
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    const int bj = bi + rem;
    const int I0 = bi * GB_, J0 = bj * GB_;
    const int rows_per = ((n + split - 1) / split + 3) / 4 * 4;
    const int r_begin = sk * rows_per, r_end = min(n, r_begin + rows_per);
    const int ci = lane & 15, kr = lane >> 4;
    d4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (d4_t){0., 0., 0., 0.};
    for (int r0 = r_begin; r0 < r_end; r0 += 4) {
        const int row = r0 + kr;
        const bool rok = row < r_end;
        double fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ca = I0 + 16 * t + ci, cb = J0 + 16 * t + ci;
            fa[t] = (rok && ca < P) ? A[(size_t)row * lda + ca] : 0.;
            fb[t] = (rok && cb < P) ? A[(size_t)row * lda + cb] : 0.;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    // D[row = kr + 4 r][col = ci] of tile (a, b)
    double *out = part + ((size_t)sk * n_blk + blk) * (GB_ * GB_);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * a + kr + 4 * r) * GB_ + 16 * b + ci] = acc[a][b][r];
}

__global__ void bf_gram_reduce_kernel(int P, int nb, int split, const double *__restrict__ part, double *__restrict__ G) {
    const int n_blk = nb * (nb + 1) / 2;
    const int blk = blockIdx.x;
    int bi = 0, rem = blk;
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    const int bj = bi + rem;
    for (int e = threadIdx.x; e < GB_ * GB_; e += blockDim.x) {
        double s = 0.;
        for (int k = 0; k < split; ++k) s += part[((size_t)k * n_blk + blk) * (GB_ * GB_) + e];
        const int i = bi * GB_ + e / GB_, j = bj * GB_ + e % GB_;
        if (i < P && j < P) {
            G[(size_t)i * P + j] = s;
            if (bi != bj) G[(size_t)j * P + i] = s;
        }
    }
}

// r = A^T B in two deterministic passes: partial sums over NSEG row segments (one thread per column and segment,
// rows in order), then the segments in order.  B (n, m) row-major, m small.
#define ATB_SEG_ 32
__global__ void bf_atb_kernel(int n, int P, int m, const double *__restrict__ A, int lda, const double *__restrict__ B,
                              double *__restrict__ part) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, sg = blockIdx.z;
    if (j >= P) return;
    const int rows = (n + ATB_SEG_ - 1) / ATB_SEG_;
    const int i0 = sg * rows, i1 = min(n, i0 + rows);
    double s = 0.;
    for (int i = i0; i < i1; ++i) s += A[(size_t)i * lda + j] * B[(size_t)i * m + c];
    part[((size_t)sg * m + c) * P + j] = s;
}
__global__ void bf_atb_reduce_kernel(int P, int m, const double *__restrict__ part, double *__restrict__ r) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= P) return;
    double s = 0.;
    for (int sg = 0; sg < ATB_SEG_; ++sg) s += part[((size_t)sg * m + c) * P + j];
    r[(size_t)j * m + c] = s;
}

static int ensure_scratch(bfhip_ctx *ctx, size_t need) {
    if (ctx->scratch_bytes < need) {
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) BF_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = NULL;
        ctx->scratch_bytes = 0;
        BF_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    return 0;
}

extern "C" int bfhip_gram(bfhip_ctx *ctx, int n, int P, int m, const double *A, int lda, const double *B, double *G, double *r) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 1 || P < 1 || m < 0 || !A || !G || lda < P || (m > 0 && (!B || !r)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_gram: invalid argument");
    const int nb = (P + GB_ - 1) / GB_;
    const int n_blk = nb * (nb + 1) / 2;
    int split = (4 * ctx->n_cu * 2 + n_blk - 1) / n_blk;  // aim at ~2 waves per SIMD
    if (split < 1) split = 1;
    if (split > 16) split = 16;
    if (split > (n + 63) / 64) split = (n + 63) / 64;
    size_t need = (size_t)split * n_blk * GB_ * GB_ * sizeof(double);
    const size_t need_atb = (size_t)ATB_SEG_ * (m > 0 ? m : 1) * P * sizeof(double);
    if (need < need_atb) need = need_atb;
    if (int rc = ensure_scratch(ctx, need)) return rc;
    double *part = (double *)ctx->scratch;
    const int waves = n_blk * split;
    hipLaunchKernelGGL(bf_gram_kernel, dim3((waves + 3) / 4), dim3(256), 0, ctx->stream, n, P, A, lda, nb, split, part);
    hipLaunchKernelGGL(bf_gram_reduce_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, P, nb, split, part, G);
    if (m > 0) {  // (the Gram partials have been consumed by the reduce kernel: the scratch is free again)
        hipLaunchKernelGGL(bf_atb_kernel, dim3((P + 127) / 128, m, ATB_SEG_), dim3(128), 0, ctx->stream, n, P, m, A, lda, B, part);
        hipLaunchKernelGGL(bf_atb_reduce_kernel, dim3((P + 127) / 128, m), dim3(128), 0, ctx->stream, P, m, part, r);
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// SPD solve: Jacobi equilibration, blocked right-looking Cholesky (NB = 64), two triangular solves.
// ---------------------------------------------------------------------------------------------------
#define NB_ 64

// two-pass equilibration (the diagonal is read before anything is scaled)
__global__ void bf_diag_kernel(int P, const double *__restrict__ G, double *__restrict__ dsc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) {
        const double gii = G[(size_t)i * P + i];
        dsc[i] = gii > 0. ? 1. / sqrt(gii) : 1.;
    }
}
__global__ void bf_scale_kernel(int P, int m, double *__restrict__ G, double *__restrict__ r, const double *__restrict__ dsc) {
    const int i = blockIdx.x;
    const double di = dsc[i];
    for (int j = threadIdx.x; j < P; j += blockDim.x) G[(size_t)i * P + j] *= di * dsc[j];
    for (int c = threadIdx.x; c < m; c += blockDim.x) r[(size_t)i * m + c] *= di;
}
__global__ void bf_unscale_kernel(int P, int m, double *__restrict__ r, const double *__restrict__ dsc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P)
        for (int c = 0; c < m; ++c) r[(size_t)i * m + c] *= dsc[i];
}

// Cholesky of the NB x NB diagonal block at (k0, k0) by ONE wave (lane = row, left-looking, block in LDS), followed
// by the inverse of the factor (forward substitution on the identity, lane = column), which turns the panel solve
// below into a small matrix product.  info = first pivot below the threshold (1-based).
__global__ __launch_bounds__(64) void bf_chol_diag_kernel(int P, int k0, double *__restrict__ G, double *__restrict__ Linv,
                                                         int *__restrict__ info) {
    __shared__ double L[NB_][NB_ + 1];
    __shared__ double X[NB_][NB_ + 1];
    const int nbk = min(NB_, P - k0);
    const int t = threadIdx.x;
    for (int i = 0; i < NB_; ++i)  // row i: one coalesced 512-byte read
        L[i][t] = (i < nbk && t < nbk) ? G[(size_t)(k0 + i) * P + k0 + t] : (i == t ? 1. : 0.);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < nbk; ++j) {
        // lane i >= j: a[i][j] - sum_{k<j} L[i][k] L[j][k]
        double acc = L[t][j];
#pragma unroll 8
        for (int k = 0; k < j; ++k) acc -= L[t][k] * L[j][k];
        double djj = bf_readlane(acc, j);
        // the matrix is equilibrated (unit diagonal): a pivot below 1e-11 puts the condition number of the normal
        // equations beyond the range in which the refinement of bfhip_lstsq_refine contracts quickly (its factor is
        // ~ P eps cond(G)): the design matrix is reported as numerically rank deficient
        if (!(djj > 1e-11)) {
            if (t == 0 && *info == 0) *info = k0 + j + 1;
            djj = 1.;
        }
        const double ljj = sqrt(djj);
        __builtin_amdgcn_wave_barrier();
        if (t == j) L[t][j] = ljj;
        else if (t > j) L[t][j] = acc / ljj;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = 0; i < NB_; ++i)
        if (i < nbk && t < nbk) G[(size_t)(k0 + i) * P + k0 + t] = t <= i ? L[i][t] : 0.;
    // X = L^-1: lane c solves L x = e_c
    for (int j = 0; j < NB_; ++j) {
        double sacc = (j == t) ? 1. : 0.;
        if (j >= t) {
#pragma unroll 8
            for (int k = t; k < j; ++k) sacc -= L[j][k] * X[k][t];
            sacc /= L[j][j];
        } else {
            sacc = 0.;
        }
        X[j][t] = sacc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < NB_; ++i) Linv[i * NB_ + t] = X[i][t];
}

// panel: rows below the diagonal block, X = A21 L11^-T = A21 (L11^-1)^T: 64-row tiles, each thread 16 outputs
__global__ __launch_bounds__(256) void bf_chol_trsm_kernel(int P, int k0, double *__restrict__ G, const double *__restrict__ Linv) {
    __shared__ double Li[NB_][NB_ + 1];
    __shared__ double At[NB_][NB_ + 1];
    const int nbk = min(NB_, P - k0);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int r0 = k0 + nbk + blockIdx.x * NB_;
    for (int i = wv; i < NB_; i += 4) {
        Li[i][lane] = Linv[i * NB_ + lane];
        const int row = r0 + i;
        At[i][lane] = (row < P && lane < nbk) ? G[(size_t)row * P + k0 + lane] : 0.;
    }
    __syncthreads();
    // thread (row = lane, columns wv*16 .. wv*16+15):  X[row][j] = sum_{k <= j} A[row][k] Linv[j][k]
    const int row = r0 + lane;
    double out[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = wv * 16 + jj;
        double sacc = 0.;
        for (int k = 0; k <= j; ++k) sacc += At[lane][k] * Li[j][k];
        out[jj] = sacc;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) At[lane][wv * 16 + jj] = out[jj];
    __syncthreads();
    for (int i = wv; i < NB_; i += 4) {
        const int rr = r0 + i;
        if (rr < P && lane < nbk) G[(size_t)rr * P + k0 + lane] = At[i][lane];
    }
    (void)row;
}

// trailing update A22 -= L21 L21^T on MFMA, lower-triangular 64 x 64 blocks, one wave per block
__global__ __launch_bounds__(256) void bf_chol_syrk_kernel(int P, int k0, double *__restrict__ G) {
    const int nbk = min(NB_, P - k0);
    const int t0 = k0 + nbk;  // first trailing row
    const int nt = P - t0;
    const int nb = (nt + GB_ - 1) / GB_;
    const int n_blk = nb * (nb + 1) / 2;
    const int lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= n_blk) return;
    int bj = 0, rem = blk;  // lower triangle: bi >= bj
    while (rem >= nb - bj) { rem -= nb - bj; ++bj; }
    const int bi = bj + rem;
    const int I0 = t0 + bi * GB_, J0 = t0 + bj * GB_;
    const int ci = lane & 15, kr = lane >> 4;
    d4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (d4_t){0., 0., 0., 0.};
    for (int kk = 0; kk < nbk; kk += 4) {
        const int col = k0 + kk + kr;
        const bool cok = kk + kr < nbk;
        double fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ra = I0 + 16 * t + ci, rb = J0 + 16 * t + ci;
            fa[t] = (cok && ra < P) ? G[(size_t)ra * P + col] : 0.;  // A operand: L21[I + i][k]
            fb[t] = (cok && rb < P) ? G[(size_t)rb * P + col] : 0.;  // B operand: L21[J + j][k]
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = I0 + 16 * a + kr + 4 * r, j = J0 + 16 * b + ci;
                if (i < P && j < P && j <= i) G[(size_t)i * P + j] -= acc[a][b][r];
            }
}

// (wave-level sums used by the residual kernel of bfhip_lstsq)
template <int CTRL>
__device__ inline double bf_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sums of N values over the 64 lanes, advanced together: DPP butterflies inside the rows of 16 lanes, then the four
// row totals by v_readlane, in a fixed order (no LDS crossbar)
template <int N>
__device__ inline void bf_wave_sum_n(double (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += bf_dpp_f64<0xB1>(v[i]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += bf_dpp_f64<0x4E>(v[i]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += bf_dpp_f64<0x141>(v[i]);  // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += bf_dpp_f64<0x140>(v[i]);  // row_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ((bf_readlane(v[i], 0) + bf_readlane(v[i], 16)) + bf_readlane(v[i], 32)) + bf_readlane(v[i], 48);
}
static int solve_spd_impl(bfhip_ctx *ctx, int P, int m, double *G, double *r, int *info, double *dsc, double *LinvAll) {
    hipStream_t st = ctx->stream;
    BF_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), st));
    hipLaunchKernelGGL(bf_diag_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, G, dsc);
    hipLaunchKernelGGL(bf_scale_kernel, dim3(P), dim3(256), 0, st, P, m, G, r, dsc);
    for (int k0 = 0; k0 < P; k0 += NB_) {
        const int nbk = P - k0 < NB_ ? P - k0 : NB_;
        double *Linv = LinvAll + (size_t)(k0 / NB_) * NB_ * NB_;  // kept: the triangular solves multiply by it
        hipLaunchKernelGGL(bf_chol_diag_kernel, dim3(1), dim3(64), 0, st, P, k0, G, Linv, info);
        const int below = P - k0 - nbk;
        if (below > 0) {
            hipLaunchKernelGGL(bf_chol_trsm_kernel, dim3((below + NB_ - 1) / NB_), dim3(256), 0, st, P, k0, G, Linv);
            const int nb = (below + GB_ - 1) / GB_;
            const int n_blk = nb * (nb + 1) / 2;
            hipLaunchKernelGGL(bf_chol_syrk_kernel, dim3((n_blk + 3) / 4), dim3(256), 0, st, P, k0, G);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Triangular solves by block columns, one launch per 64-column panel and sweep (the single-workgroup sweep this
// replaces read the whole factor through one CU: 1.1 ms per sweep at P = 2145).  Every workgroup of a launch forms the
// panel's 64 unknowns itself, by a product with the inverse of the diagonal block kept from the factorisation, and
// then takes the panel's contribution out of its own 64 rows (forward) or columns (backward) of the right-hand side.
// Sums run in index order: the result does not depend on the grid.
// ---------------------------------------------------------------------------------------------------
// forward, panel at k0: y_k = L_kk^-1 r_k;  r_i -= L_ik y_k for the rows i below the panel
__global__ __launch_bounds__(256) void bf_trsv_fwd_kernel(int P, int m, int k0, const double *__restrict__ L,
                                                          const double *__restrict__ Linv, double *__restrict__ r,
                                                          double *__restrict__ y) {
    __shared__ double T[NB_][NB_ + 1];
    __shared__ double rk[NB_], yk[NB_];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nbk = min(NB_, P - k0);
    const int i0 = k0 + nbk + blockIdx.x * NB_;
    for (int q = 0; q < m; ++q) {
        for (int i = wv; i < NB_; i += 4) T[i][lane] = Linv[i * NB_ + lane];
        if (t < NB_) rk[t] = t < nbk ? r[(size_t)(k0 + t) * m + q] : 0.;
        __syncthreads();
        if (t < NB_) {
            double acc = 0.;
            for (int c = 0; c <= t; ++c) acc += T[t][c] * rk[c];
            yk[t] = acc;
            if (blockIdx.x == 0 && t < nbk) y[(size_t)(k0 + t) * m + q] = acc;
        }
        __syncthreads();
        if (i0 < P) {
            for (int i = wv; i < NB_; i += 4) T[i][lane] = (i0 + i < P && lane < nbk) ? L[(size_t)(i0 + i) * P + k0 + lane] : 0.;
            __syncthreads();
            if (t < NB_ && i0 + t < P) {
                double acc = 0.;
                for (int c = 0; c < nbk; ++c) acc += T[t][c] * yk[c];
                r[(size_t)(i0 + t) * m + q] -= acc;
            }
        }
        __syncthreads();
    }
}
// backward, panel at k0: x_k = L_kk^-T y_k;  y_j -= L_kj^T x_k for the columns j left of the panel
__global__ __launch_bounds__(256) void bf_trsv_bwd_kernel(int P, int m, int k0, const double *__restrict__ L,
                                                          const double *__restrict__ Linv, double *__restrict__ y,
                                                          double *__restrict__ x) {
    __shared__ double T[NB_][NB_ + 1];
    __shared__ double yk[NB_], xk[NB_];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nbk = min(NB_, P - k0);
    const int j0 = blockIdx.x * NB_;
    for (int q = 0; q < m; ++q) {
        for (int i = wv; i < NB_; i += 4) T[i][lane] = Linv[i * NB_ + lane];
        if (t < NB_) yk[t] = t < nbk ? y[(size_t)(k0 + t) * m + q] : 0.;
        __syncthreads();
        if (t < NB_) {
            double acc = 0.;
            for (int i = t; i < NB_; ++i) acc += T[i][t] * yk[i];
            xk[t] = acc;
            if (blockIdx.x == 0 && t < nbk) x[(size_t)(k0 + t) * m + q] = acc;
        }
        __syncthreads();
        if (j0 < k0) {
            for (int i = wv; i < NB_; i += 4) T[i][lane] = i < nbk ? L[(size_t)(k0 + i) * P + j0 + lane] : 0.;
            __syncthreads();
            if (t < NB_) {
                double acc = 0.;
                for (int i = 0; i < nbk; ++i) acc += T[i][t] * xk[i];
                y[(size_t)(j0 + t) * m + q] -= acc;
            }
        }
        __syncthreads();
    }
}

// x = (D L L^T D)^-1 r with the kept factor L, the inverses of its diagonal blocks and the scales D = diag(dsc): r is
// overwritten by x; ybuf (P, m) is work space
static int chol_apply(bfhip_ctx *ctx, int P, int m, const double *L, const double *dsc, const double *LinvAll, double *ybuf,
                      double *r, bool scale_in) {
    hipStream_t st = ctx->stream;
    if (scale_in) hipLaunchKernelGGL(bf_unscale_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, m, r, dsc);
    const int nblk = (P + NB_ - 1) / NB_;
    for (int kb = 0; kb < nblk; ++kb) {
        const int k0 = kb * NB_, nbk = P - k0 < NB_ ? P - k0 : NB_, below = P - k0 - nbk;
        const int grid = below > 0 ? (below + NB_ - 1) / NB_ : 1;
        hipLaunchKernelGGL(bf_trsv_fwd_kernel, dim3(grid), dim3(256), 0, st, P, m, k0, L, LinvAll + (size_t)kb * NB_ * NB_, r, ybuf);
    }
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * NB_;
        const int grid = kb > 0 ? kb : 1;
        hipLaunchKernelGGL(bf_trsv_bwd_kernel, dim3(grid), dim3(256), 0, st, P, m, k0, L, LinvAll + (size_t)kb * NB_ * NB_, ybuf, r);
    }
    hipLaunchKernelGGL(bf_unscale_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, m, r, dsc);
    return 0;
}

extern "C" int bfhip_solve_spd(bfhip_ctx *ctx, int P, int m, double *G, double *r, int *info) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || P < 1 || m < 1 || !G || !r || !info) return bf_set_error(BFHIP_ERR_ARG, "bfhip_solve_spd: invalid argument");
    const size_t n_inv = (size_t)((P + NB_ - 1) / NB_) * NB_ * NB_;
    if (int rc = ensure_scratch(ctx, ((size_t)P + n_inv + (size_t)P * m) * sizeof(double))) return rc;
    double *dsc = (double *)ctx->scratch, *LinvAll = dsc + P, *ybuf = LinvAll + n_inv;
    if (int rc = solve_spd_impl(ctx, P, m, G, r, info, dsc, LinvAll)) return rc;
    if (int rc = chol_apply(ctx, P, m, G, dsc, LinvAll, ybuf, r, false)) return rc;
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// least squares = normal equations + refinement on the TRUE residual (corrected semi-normal equations):
//   c_0 = G^-1 A^T B,   c_{k+1} = c_k + G^-1 A^T (B - A c_k)
// The fixed point is the least-squares solution whatever the rounding of G and of its factor; the error contracts by
// ~ P eps cond(G) per step, so that two steps bring the coefficients from the cond(A)^2 eps of plain normal equations
// to the cond(A) eps of an orthogonal factorisation (LAPACK gelsd, modules/poly.py:570) for every design the pivot
// threshold lets through.
// ---------------------------------------------------------------------------------------------------
// S = B - A c, one wave per row (coalesced 512-byte reads), fixed summation order
__global__ __launch_bounds__(256) void bf_resid_kernel(int n, int P, int m, const double *__restrict__ A, int lda,
                                                       const double *__restrict__ B, const double *__restrict__ c,
                                                       double *__restrict__ S) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    for (int q = 0; q < m; ++q) {
        double v[1] = {0.};
        for (int k = lane; k < P; k += 64) v[0] += A[(size_t)row * lda + k] * c[(size_t)k * m + q];
        bf_wave_sum_n<1>(v);
        if (lane == 0) S[(size_t)row * m + q] = B[(size_t)row * m + q] - v[0];
    }
}
__global__ void bf_axpy_kernel(int n, double *__restrict__ x, const double *__restrict__ dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += dx[i];
}

extern "C" int bfhip_lstsq(bfhip_ctx *ctx, int n, int P, int m, const double *A, int lda, const double *B, double *G, double *c,
                           int n_refine, double *work, int *info) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || n < 1 || P < 1 || m < 1 || !A || !B || !G || !c || !info || lda < P || n_refine < 0 || (n_refine > 0 && !work))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_lstsq: invalid argument");
    if (int rc = bfhip_gram(ctx, n, P, m, A, lda, B, G, c)) return rc;
    // (the Gram scratch is free again; the scales and the block inverse live behind the A^T S partials of the refinement)
    const size_t n_atb = (size_t)ATB_SEG_ * m * P, n_inv = (size_t)((P + NB_ - 1) / NB_) * NB_ * NB_;
    if (int rc = ensure_scratch(ctx, (n_atb + P + n_inv + (size_t)P * m) * sizeof(double))) return rc;
    double *part = (double *)ctx->scratch, *dsc = part + n_atb, *LinvAll = dsc + P, *ybuf = LinvAll + n_inv;
    if (int rc = solve_spd_impl(ctx, P, m, G, c, info, dsc, LinvAll)) return rc;
    if (int rc = chol_apply(ctx, P, m, G, dsc, LinvAll, ybuf, c, false)) return rc;
    double *S = work, *dc = work + (size_t)n * m;
    hipStream_t st = ctx->stream;
    for (int it = 0; it < n_refine; ++it) {
        hipLaunchKernelGGL(bf_resid_kernel, dim3((n + 3) / 4), dim3(256), 0, st, n, P, m, A, lda, B, c, S);
        hipLaunchKernelGGL(bf_atb_kernel, dim3((P + 127) / 128, m, ATB_SEG_), dim3(128), 0, st, n, P, m, A, lda, S, part);
        hipLaunchKernelGGL(bf_atb_reduce_kernel, dim3((P + 127) / 128, m), dim3(128), 0, st, P, m, part, dc);
        if (int rc = chol_apply(ctx, P, m, G, dsc, LinvAll, ybuf, dc, true)) return rc;
        hipLaunchKernelGGL(bf_axpy_kernel, dim3((P * m + 255) / 256), dim3(256), 0, st, P * m, c, dc);
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
