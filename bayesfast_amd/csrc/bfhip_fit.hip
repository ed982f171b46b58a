// bfhip_fit.hip -- surrogate fit on device: design-matrix blocks, normal equations on FP64 MFMA, SPD solve.
//
// PolyModel.fit (modules/poly.py:505-589) builds A = [1 | x | quadratic | cubic-2 | cubic-3] per output and
// calls scipy.linalg.lstsq (LAPACK gelsd) on it, once per output.  Here the design blocks are written by a
// memory-bound kernel, G = A^T A is a genuine dense contraction and runs on v_mfma_f64_16x16x4_f64, and the
// P x P system is solved by a blocked Cholesky with Jacobi equilibration (all outputs share one
// factorisation).
#include <vector>
#include "bfhip_common.h"

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

// r = A^T B: one thread per column of A, rows in order (deterministic); B (n, m) row-major, m small
__global__ void bf_atb_kernel(int n, int P, int m, const double *__restrict__ A, int lda, const double *__restrict__ B,
                              double *__restrict__ r) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= P) return;
    double s = 0.;
    for (int i = 0; i < n; ++i) s += A[(size_t)i * lda + j] * B[(size_t)i * m + c];
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
    if (!ctx || n < 1 || P < 1 || m < 0 || !A || !G || lda < P || (m > 0 && (!B || !r)))
        return bf_set_error(BFHIP_ERR_ARG, "bfhip_gram: invalid argument");
    const int nb = (P + GB_ - 1) / GB_;
    const int n_blk = nb * (nb + 1) / 2;
    int split = (4 * ctx->n_cu * 2 + n_blk - 1) / n_blk;  // aim at ~2 waves per SIMD
    if (split < 1) split = 1;
    if (split > 16) split = 16;
    if (split > (n + 63) / 64) split = (n + 63) / 64;
    const size_t need = (size_t)split * n_blk * GB_ * GB_ * sizeof(double);
    if (int rc = ensure_scratch(ctx, need)) return rc;
    double *part = (double *)ctx->scratch;
    const int waves = n_blk * split;
    hipLaunchKernelGGL(bf_gram_kernel, dim3((waves + 3) / 4), dim3(256), 0, ctx->stream, n, P, A, lda, nb, split, part);
    hipLaunchKernelGGL(bf_gram_reduce_kernel, dim3(n_blk), dim3(256), 0, ctx->stream, P, nb, split, part, G);
    if (m > 0) hipLaunchKernelGGL(bf_atb_kernel, dim3((P + 127) / 128, m), dim3(128), 0, ctx->stream, n, P, m, A, lda, B, r);
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

// Cholesky of the NB x NB diagonal block at (k0, k0), in LDS; info = first non-positive pivot (1-based)
__global__ void bf_chol_diag_kernel(int P, int k0, double *__restrict__ G, int *__restrict__ info) {
    __shared__ double L[NB_][NB_ + 1];
    const int nbk = min(NB_, P - k0);
    const int t = threadIdx.x;
    for (int e = t; e < NB_ * NB_; e += blockDim.x) {
        const int i = e / NB_, j = e % NB_;
        L[i][j] = (i < nbk && j < nbk) ? G[(size_t)(k0 + i) * P + k0 + j] : (i == j ? 1. : 0.);
    }
    __syncthreads();
    for (int j = 0; j < nbk; ++j) {
        if (t == 0) {
            const double djj = L[j][j];
            // the matrix is equilibrated (unit diagonal): a pivot below 1e-13 means a numerically rank
            // deficient design matrix (condition number of the normal equations beyond double precision)
            if (!(djj > 1e-13)) {
                if (*info == 0) *info = k0 + j + 1;
                L[j][j] = 1.;
            } else {
                L[j][j] = sqrt(djj);
            }
        }
        __syncthreads();
        const double ljj = L[j][j];
        for (int i = j + 1 + t; i < nbk; i += blockDim.x) L[i][j] /= ljj;
        __syncthreads();
        // trailing update of the lower triangle
        for (int e = t; e < (nbk - j - 1) * (nbk - j - 1); e += blockDim.x) {
            const int i = j + 1 + e / (nbk - j - 1), c = j + 1 + e % (nbk - j - 1);
            if (c <= i) L[i][c] -= L[i][j] * L[c][j];
        }
        __syncthreads();
    }
    for (int e = t; e < NB_ * NB_; e += blockDim.x) {
        const int i = e / NB_, j = e % NB_;
        if (i < nbk && j < nbk) G[(size_t)(k0 + i) * P + k0 + j] = j <= i ? L[i][j] : 0.;
    }
}

// panel: rows below the diagonal block, X L11^T = A21 (one thread per row, L11 in LDS)
__global__ void bf_chol_trsm_kernel(int P, int k0, double *__restrict__ G) {
    __shared__ double L[NB_][NB_ + 1];
    const int nbk = min(NB_, P - k0);
    for (int e = threadIdx.x; e < NB_ * NB_; e += blockDim.x) {
        const int i = e / NB_, j = e % NB_;
        L[i][j] = (i < nbk && j < nbk) ? G[(size_t)(k0 + i) * P + k0 + j] : 0.;
    }
    __syncthreads();
    const int row = k0 + nbk + blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= P) return;
    double *a = G + (size_t)row * P + k0;
    double xr[NB_];
    for (int j = 0; j < nbk; ++j) {
        double s = a[j];
        for (int c = 0; c < j; ++c) s -= xr[c] * L[j][c];
        xr[j] = s / L[j][j];
    }
    for (int j = 0; j < nbk; ++j) a[j] = xr[j];
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

// forward substitution L y = r over one diagonal block + update of the rows below (one workgroup)
__global__ void bf_trsv_fwd_kernel(int P, int m, int k0, const double *__restrict__ G, double *__restrict__ r) {
    __shared__ double L[NB_][NB_ + 1];
    __shared__ double y[NB_];
    const int nbk = min(NB_, P - k0);
    const int t = threadIdx.x;
    for (int e = t; e < NB_ * NB_; e += blockDim.x) {
        const int i = e / NB_, j = e % NB_;
        L[i][j] = (i < nbk && j < nbk) ? G[(size_t)(k0 + i) * P + k0 + j] : 0.;
    }
    for (int c = 0; c < m; ++c) {
        __syncthreads();
        if (t < nbk) y[t] = r[(size_t)(k0 + t) * m + c];
        __syncthreads();
        for (int j = 0; j < nbk; ++j) {
            if (t == 0) y[j] /= L[j][j];
            __syncthreads();
            if (t > j && t < nbk) y[t] -= L[t][j] * y[j];
            __syncthreads();
        }
        if (t < nbk) r[(size_t)(k0 + t) * m + c] = y[t];
        // rows below: r_i -= L[i, k0:k0+nbk] . y
        for (int i = k0 + nbk + t; i < P; i += blockDim.x) {
            const double *li = G + (size_t)i * P + k0;
            double s = 0.;
            for (int j = 0; j < nbk; ++j) s += li[j] * y[j];
            r[(size_t)i * m + c] -= s;
        }
    }
}

// backward substitution L^T c = y, blocks from the bottom up
__global__ void bf_trsv_bwd_kernel(int P, int m, int k0, const double *__restrict__ G, double *__restrict__ r) {
    __shared__ double L[NB_][NB_ + 1];
    __shared__ double y[NB_];
    const int nbk = min(NB_, P - k0);
    const int t = threadIdx.x;
    for (int e = t; e < NB_ * NB_; e += blockDim.x) {
        const int i = e / NB_, j = e % NB_;
        L[i][j] = (i < nbk && j < nbk) ? G[(size_t)(k0 + i) * P + k0 + j] : 0.;
    }
    for (int c = 0; c < m; ++c) {
        __syncthreads();
        // y_j = r_j - sum_{i >= k0 + nbk} L[i][k0 + j] x_i   (x below is final already)
        if (t < nbk) {
            double s = r[(size_t)(k0 + t) * m + c];
            for (int i = k0 + nbk; i < P; ++i) s -= G[(size_t)i * P + k0 + t] * r[(size_t)i * m + c];
            y[t] = s;
        }
        __syncthreads();
        for (int j = nbk - 1; j >= 0; --j) {
            if (t == 0) y[j] /= L[j][j];
            __syncthreads();
            if (t < j) y[t] -= L[j][t] * y[j];
            __syncthreads();
        }
        if (t < nbk) r[(size_t)(k0 + t) * m + c] = y[t];
    }
}

extern "C" int bfhip_solve_spd(bfhip_ctx *ctx, int P, int m, double *G, double *r, int *info) {
    if (!ctx || P < 1 || m < 1 || !G || !r || !info) return bf_set_error(BFHIP_ERR_ARG, "bfhip_solve_spd: invalid argument");
    if (int rc = ensure_scratch(ctx, (size_t)P * sizeof(double))) return rc;
    double *dsc = (double *)ctx->scratch;
    hipStream_t st = ctx->stream;
    BF_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), st));
    hipLaunchKernelGGL(bf_diag_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, G, dsc);
    hipLaunchKernelGGL(bf_scale_kernel, dim3(P), dim3(256), 0, st, P, m, G, r, dsc);
    for (int k0 = 0; k0 < P; k0 += NB_) {
        const int nbk = P - k0 < NB_ ? P - k0 : NB_;
        hipLaunchKernelGGL(bf_chol_diag_kernel, dim3(1), dim3(256), 0, st, P, k0, G, info);
        const int below = P - k0 - nbk;
        if (below > 0) {
            hipLaunchKernelGGL(bf_chol_trsm_kernel, dim3((below + 63) / 64), dim3(64), 0, st, P, k0, G);
            const int nb = (below + GB_ - 1) / GB_;
            const int n_blk = nb * (nb + 1) / 2;
            hipLaunchKernelGGL(bf_chol_syrk_kernel, dim3((n_blk + 3) / 4), dim3(256), 0, st, P, k0, G);
        }
    }
    for (int k0 = 0; k0 < P; k0 += NB_) hipLaunchKernelGGL(bf_trsv_fwd_kernel, dim3(1), dim3(256), 0, st, P, m, k0, G, r);
    for (int k0 = (P - 1) / NB_ * NB_; k0 >= 0; k0 -= NB_)
        hipLaunchKernelGGL(bf_trsv_bwd_kernel, dim3(1), dim3(256), 0, st, P, m, k0, G, r);
    hipLaunchKernelGGL(bf_unscale_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, m, r, dsc);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
