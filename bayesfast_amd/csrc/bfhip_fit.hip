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
    // Software pipeline over steps of four rows: the operands of the step after next are on their way while a step's 16 matrix
    // instructions run (three register sets taken in turn by position in the loop body: no moves that would wait for the
    // loads).  A diagonal block (bi == bj) has one operand set.  Column guards are hoisted: only the last block column of G
    // is ragged.
    const bool diag = bi == bj;
    const double *pa[4], *pb[4];
    bool oka[4], okb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ca = I0 + 16 * t + ci, cb = J0 + 16 * t + ci;
        oka[t] = ca < P;
        okb[t] = cb < P;
        pa[t] = A + (oka[t] ? ca : 0);
        pb[t] = A + (okb[t] ? cb : 0);
    }
    auto fetch = [&](int r0, double (&fa)[4], double (&fb)[4]) {
        const int row = r0 + kr;
        const bool rok = row < r_end;
        const size_t ro = (size_t)(rok ? row : r_begin) * lda;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double va = pa[t][ro];
            fa[t] = (rok && oka[t]) ? va : 0.;
        }
        if (!diag) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double vb = pb[t][ro];
                fb[t] = (rok && okb[t]) ? vb : 0.;
            }
        }
    };
    auto run = [&](const double (&fa)[4], const double (&fb)[4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], diag ? fa[b] : fb[b], acc[a][b], 0, 0, 0);
    };
    double fa0[4], fb0[4], fa1[4], fb1[4], fa2[4], fb2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fb0[t] = fb1[t] = fb2[t] = 0.;
    if (r_begin < r_end) {
        fetch(r_begin, fa0, fb0);
        fetch(r_begin + 4, fa1, fb1);   // (rows past r_end read as zeros)
        for (int r0 = r_begin; r0 < r_end; r0 += 12) {
            fetch(r0 + 8, fa2, fb2);
            run(fa0, fb0);
            if (r0 + 4 < r_end) {
                fetch(r0 + 12, fa0, fb0);
                run(fa1, fb1);
            }
            if (r0 + 8 < r_end) {
                fetch(r0 + 16, fa1, fb1);
                run(fa2, fb2);
            }
        }
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

// The same partial sums from a workgroup of four waves on a 128 x 128 block (BI <= BJ in units of 128 columns): wave (wi, wj) owns
// the 64 x 64 sub-block (2 BI + wi, 2 BJ + wj) with the accumulation order of bf_gram_kernel (steps of four rows, ascending), so its
// partials are bf_gram_kernel's bit for bit -- but the two 128-column panels of a stage of 16 rows go through LDS once for the four
// waves: half the bytes per multiply-add from L2 / HBM.  (The design matrix of config 5's fit is 1.35 GB; with one wave per 64 x 64
// block the kernel ran at 47 % of the FP64 matrix rate, waiting for its operands: profiles/r05b_fit_profile.txt.)
#define G2_R 16        // rows per stage
#define G2_LD 272      // LDS row stride in doubles: rows kr and kr + 1 of a k-step start 32 banks apart
__global__ __launch_bounds__(256, 2) void bf_gram128_kernel(int n, int P, const double *__restrict__ A, int lda, int nb,
                                                            int split, double *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double g2_lds[];   // [2][G2_R][G2_LD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wi = w >> 1, wj = w & 1;
    const int nb2 = (nb + 1) / 2, n_blk = nb * (nb + 1) / 2;
    const int blk2 = blockIdx.x / split, sk = blockIdx.x % split;
    int BI = 0, rem = blk2;
    while (rem >= nb2 - BI) { rem -= nb2 - BI; ++BI; }
    const int BJ = BI + rem;
    const bool diag2 = BI == BJ;
    const int I0 = BI * 128, J0 = BJ * 128;
    const int rows_per = ((n + split - 1) / split + 3) / 4 * 4;
    const int r_begin = sk * rows_per, r_end = min(n, r_begin + rows_per);
    const int bi = 2 * BI + wi, bj = 2 * BJ + wj;
    const bool mine = bi < nb && bj < nb && bi <= bj;   // (the mirror sub-block of a diagonal block and the ragged last column: idle)
    const bool diag = bi == bj;
    const int ci = lane & 15, kr = lane >> 4;
    d4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (d4_t){0., 0., 0., 0.};
    // staging: thread t moves elements e = t + 256 j (j < 16) of a stage, row e / 256, column e % 256 (0-127: panel I, 128-255: J)
    const int scol = tid, ncol = diag2 ? 128 : 256;
    const int gcol = scol < 128 ? I0 + scol : J0 + scol - 128;
    const bool cok = scol < ncol && gcol < P;
    const double *gp = A + (cok ? gcol : 0);
    double st[G2_R];
    auto gload = [&](int r0) {
        if (r0 + G2_R <= r_end) {   // (a full stage: sixteen loads on their way together -- with the row guard below the compiler
            const double *rp = gp + (size_t)r0 * lda;   //  branches on the uniform condition and waits for every load on its own)
#pragma unroll
            for (int j = 0; j < G2_R; ++j) st[j] = rp[(size_t)j * lda];   // (columns past P: masked when the stage is stored)
        } else {
#pragma unroll
            for (int j = 0; j < G2_R; ++j) {
                const int row = r0 + j;
                const bool rok = row < r_end;
                const double v = gp[(size_t)(rok ? row : r_begin) * lda];
                st[j] = (rok && cok) ? v : 0.;
            }
        }
    };
    auto lstore = [&](int buf) {
        double *dst = g2_lds + (size_t)buf * G2_R * G2_LD + scol;
        if (scol < ncol) {
#pragma unroll
            for (int j = 0; j < G2_R; ++j) dst[j * G2_LD] = cok ? st[j] : 0.;
        }
    };
    const int ao = wi * 64 + ci, bo = (diag2 ? 0 : 128) + wj * 64 + ci;
    auto compute = [&](int buf) {
        const double *src = g2_lds + (size_t)buf * G2_R * G2_LD;
#pragma unroll
        for (int k = 0; k < G2_R / 4; ++k) {
            double fa[4], fb[4];
            const double *rowp = src + (4 * k + kr) * G2_LD;
#pragma unroll
            for (int t = 0; t < 4; ++t) { fa[t] = rowp[ao + 16 * t]; fb[t] = rowp[bo + 16 * t]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], diag ? fa[b] : fb[b], acc[a][b], 0, 0, 0);
        }
    };
    if (r_begin < r_end) {
        gload(r_begin);
        lstore(0);
        __syncthreads();
        int buf = 0;
        for (int r0 = r_begin; r0 < r_end; r0 += G2_R, buf ^= 1) {
            const bool more = r0 + G2_R < r_end;
            if (more) gload(r0 + G2_R);
            if (mine) compute(buf);
            if (more) lstore(buf ^ 1);
            __syncthreads();
        }
    }
    if (!mine) return;
    const int blk = bi * nb - bi * (bi - 1) / 2 + (bj - bi);
    double *out = part + ((size_t)sk * n_blk + blk) * (GB_ * GB_);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * a + kr + 4 * r) * GB_ + 16 * b + ci] = acc[a][b][r];
}

// One workgroup per 64 x 64 block: the split-K partials are added in part order (fixed: bitwise reproducible), the block is
// written row by row and, for an off-diagonal block, its transpose row by row too -- through an LDS tile, so that both writes
// are coalesced (the mirror block written element by element down a column was 3/4 of this kernel's time).
__global__ __launch_bounds__(256) void bf_gram_reduce_kernel(int P, int nb, int split, const double *__restrict__ part, double *__restrict__ G) {
    __shared__ double T[GB_][GB_ + 1];
    const int n_blk = nb * (nb + 1) / 2;
    const int blk = blockIdx.x;
    int bi = 0, rem = blk;
    while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
    const int bj = bi + rem;
    const double *pp = part + (size_t)blk * (GB_ * GB_);
    const size_t stride = (size_t)n_blk * (GB_ * GB_);
    // thread t owns elements e = t + 256 k (k < 16): the sums of its 16 elements advance together over the parts
    double sacc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sacc[k] = 0.;
    for (int q = 0; q < split; ++q) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sacc[k] += pp[(size_t)q * stride + threadIdx.x + 256 * k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = threadIdx.x + 256 * k, r = e / GB_, c = e % GB_;
        const int i = bi * GB_ + r, j = bj * GB_ + c;
        if (i < P && j < P) G[(size_t)i * P + j] = sacc[k];
        T[r][c] = sacc[k];
    }
    if (bi == bj) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = threadIdx.x + 256 * k, r = e / GB_, c = e % GB_;   // element (r, c) of the mirror block = T[c][r]
        const int i = bj * GB_ + r, j = bi * GB_ + c;
        if (i < P && j < P) G[(size_t)i * P + j] = T[c][r];
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
    // split-K: a wave holds a 64 x 64 block's 128 accumulator registers, so two waves fit a SIMD and the chip has
    // 8 n_cu slots; the waves run in rounds of that many, a round lasting as long as one wave's rows.  The split that
    // minimises rounds x rows per wave (2380 waves of 1073 rows on 2048 slots would be two rounds for 1.16 rounds of work)
    int split = 1;
    {
        const long slots = 8L * ctx->n_cu;
        long best = -1;
        const int smax = (n + 63) / 64 < 16 ? (n + 63) / 64 : 16;
        for (int sp = 1; sp <= (smax > 1 ? smax : 1); ++sp) {
            const long rows = ((n + sp - 1) / sp + 3) / 4 * 4, rounds = ((long)n_blk * sp + slots - 1) / slots;
            const long cost = rounds * (rows + 64);   // (+ a wave's start-up and write-out)
            if (best < 0 || cost < best) { best = cost; split = sp; }
        }
    }
    size_t need = (size_t)split * n_blk * GB_ * GB_ * sizeof(double);
    const size_t need_atb = (size_t)ATB_SEG_ * (m > 0 ? m : 1) * P * sizeof(double);
    if (need < need_atb) need = need_atb;
    if (int rc = ensure_scratch(ctx, need)) return rc;
    double *part = (double *)ctx->scratch;
    const int waves = n_blk * split;
    if (nb >= 8 && !bf_tune().gram_one_wave) {   // (below, the 64 x 64 blocks of a small Gram matrix fill the chip better one wave each)
        const int nb2 = (nb + 1) / 2;
        const size_t lds = (size_t)2 * G2_R * G2_LD * sizeof(double);
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)bf_gram128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bf_gram128_kernel, dim3(nb2 * (nb2 + 1) / 2 * split), dim3(256), lds, ctx->stream, n, P, A, lda, nb, split, part);
    } else
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

// ---------------------------------------------------------------------------------------------------
// The factorisation: one launch per 64-column panel.  Panel k's launch has one workgroup per 64 x 64 tile (i >= j > k)
// of the trailing matrix; a workgroup forms the two blocks of the panel it needs itself, L_ik = A_ik L_kk^-T and
// L_jk, as products with the kept inverse of the diagonal block (MFMA), takes L_ik L_jk^T out of its tile, and -- the
// workgroup of tile (k+1, k+1) only -- factors the next diagonal block and inverts the factor, so that the next
// launch starts from L_kk^-1.  Nothing a launch reads is written by it: L_ik goes, transposed, into the UPPER triangle
// of G (block (k, i)), which the factorisation never reads; the triangular solves take it from there.
// The chain per panel is (update, factor 64 x 64, invert): the 64 x 64 step is what bounds the whole factorisation, so
// it runs in registers (lane = row, 64 columns in 64 registers, rank-1 updates through v_readlane) on two waves:
// wave 0 factors, wave 1 inverts one column behind it, fed through LDS.
// ---------------------------------------------------------------------------------------------------
#define LDP_ (NB_ + 2)  // (even: pairs of a row are 16-byte aligned)

// S[64][66]: the block (lower triangle; identity beyond nbk); overwritten by the factor, row-major.  Lc[64][66]: the
// factor by columns (Lc[j][t] = L[t][j]).  Called by waves 0 and 1 of the workgroup; G gets L_kk (zeros above the
// diagonal), Linv its inverse.
//   wave 0, lane = row t, a[c] = A[t][c]: at step j the column is scaled, written to LDS in both layouts, and taken
//           out of the columns to its right with its own entries L[c][j] read back as LDS broadcasts (two per read);
//   wave 1, lane = column c of X = L^-1, x[j] = X[j][c]: row j of L X = I as soon as row j of L is complete (after
//           wave 0's step j): x[j] = (delta_jc - sum_{k<j} L[j][k] x[k]) / L[j][j], the row read as LDS broadcasts.
__device__ __forceinline__ void bf_chol64_two_waves(double (*S)[LDP_], double (*Lc)[LDP_], int *ready, int P, int k0, int nbk,
                                                    double *__restrict__ G, double *__restrict__ Linv, int *__restrict__ info) {
    const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv == 0) {
        double a[NB_];
#pragma unroll
        for (int c = 0; c < NB_; ++c) a[c] = S[t][c];
        int bad = 0;
        // one basic block for all 64 steps (no branch inside): the pivot chain of step j+1 -- v_readlane, 1 / sqrt, the
        // scaled column -- only needs column j+1, which is updated first and through v_readlane (no LDS round trip), so
        // the scheduler can run it under the rest of step j's updates
#pragma unroll
        for (int j = 0; j < NB_; ++j) {
            double djj = bf_readlane(a[j], j);
            // the matrix is equilibrated (unit diagonal): a pivot below 1e-11 puts the condition number of the normal
            // equations beyond the range in which the refinement of bfhip_lstsq contracts quickly (its factor is
            // ~ P eps cond(G)): the design matrix is reported as numerically rank deficient (first such pivot, 1-based)
            const bool ok = djj > 1e-11;
            bad = (!ok && bad == 0) ? j + 1 : bad;
            djj = ok ? djj : 1.;
            const double rl = rsqrt(djj), ljj = djj * rl;
            int tt = t;
            asm volatile("" : "+v"(tt));  // (or the lane masks of all 64 steps are formed up front and spilled)
            const double lj = tt == j ? ljj : (tt > j ? a[j] * rl : 0.);
            Lc[j][t] = lj;
            S[t][j] = lj;
            if (t == 0) Lc[j][NB_] = rl;  // (the padding column: 1 / L_jj for wave 1)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (t == 0) __hip_atomic_store(ready, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            // (the pins keep the compiler from deferring the updates of the columns it does not need yet: it would
            // hold every broadcast of every step in registers and spill 2000 of them)
            if (j + 1 < NB_) {
                a[j + 1] -= lj * bf_readlane(lj, j + 1);
                asm volatile("" : "+v"(a[j + 1]));
            }
            if ((j & 1) && j + 2 < NB_) a[j + 2] -= lj * Lc[j][j + 2];  // (pairs start at an even column)
#pragma unroll
            for (int c = (j + 3) & ~1; c < NB_; c += 2) {
                const d2_t l2 = *(const d2_t *)&Lc[j][c];
                a[c] -= lj * l2.x;
                a[c + 1] -= lj * l2.y;
                if ((c & 15) == 14) {
#pragma unroll
                    for (int cc = c & ~15; cc <= c; cc += 2)
                        if (cc > j + 1) asm volatile("" : "+v"(a[cc]), "+v"(a[cc + 1]));
                }
            }
        }
        if (bad && t == 0) atomicCAS(info, 0, k0 + bad);
        for (int i = 0; i < nbk; ++i)
            if (t < nbk) G[(size_t)(k0 + i) * P + k0 + t] = S[i][t];  // (zeros above the diagonal: lj of the lanes t < j)
    } else if (wv == 1) {
        double x[NB_];
#pragma unroll
        for (int j = 0; j < NB_; ++j) {
            while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= j) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int tt = t;
            asm volatile("" : "+v"(tt));
            double sm[4] = {tt == j ? 1. : 0., 0., 0., 0.};
#pragma unroll
            for (int k = 0; k + 1 < j; k += 2) {
                const d2_t l2 = *(const d2_t *)&S[j][k];
                sm[(k >> 1) & 1] -= l2.x * x[k];
                sm[2 + ((k >> 1) & 1)] -= l2.y * x[k + 1];
            }
            if (j & 1) sm[0] -= S[j][j - 1] * x[j - 1];
            x[j] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) * Lc[j][NB_];
            asm volatile("" : "+v"(x[j]));
        }
        // (stored at the end: a store inside the loop would make every step's acquire fence wait for it)
#pragma unroll
        for (int j = 0; j < NB_; ++j) Linv[j * NB_ + t] = x[j];
    }
}

// the first diagonal block (no panel before it)
__global__ __launch_bounds__(256) void bf_chol_first_kernel(int P, double *__restrict__ G, double *__restrict__ Linv,
                                                          int *__restrict__ info) {
    __shared__ __attribute__((aligned(16))) double R0[NB_][LDP_];
    __shared__ __attribute__((aligned(16))) double R1[NB_][LDP_];
    __shared__ int ready;
    const int nbk = min(NB_, P);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < NB_; i += 4) R1[i][lane] = (i < nbk && lane < nbk) ? G[(size_t)i * P + lane] : (i == lane ? 1. : 0.);
    if (threadIdx.x == 0) ready = 0;
    __syncthreads();
    bf_chol64_two_waves(R1, R0, &ready, P, 0, nbk, G, Linv, info);
}

// panel k0 (a full 64 columns: there are rows below it): see above.
// np = 2: TWO panels in one pass over the trailing matrix (k0 and k0 + 64; the trailing blocks start at k0 + 128): every panel reads and
// writes the whole trailing matrix, 60 GB over the factorisation at P = 8385, and with the update of panel k0 delayed until panel
// k0 + 64 is there too the pass is taken half as often (measured: the 128-d fit 53.4 -> 52.0 ms -- the launches are bound by each
// workgroup's chain of dependent phases more than by the traffic).  The block column of the second panel is brought up to
// date first, by a launch of this kernel with col_only = 1 (one workgroup per block of that column: update with panel k0, L_ik into the
// upper triangle, and the column's diagonal block factored and inverted), which is what the second panel's L_i,k+1 is formed from here.
__global__ __launch_bounds__(256) void bf_chol_panel_kernel(int P, int k0, int np, int col_only, double *__restrict__ G,
                                                          double *__restrict__ LinvAll, int *__restrict__ info) {
    __shared__ __attribute__((aligned(16))) double R0[NB_][LDP_];
    __shared__ __attribute__((aligned(16))) double R1[NB_][LDP_];
    __shared__ int ready;
    const int t0 = k0 + np * NB_;  // first trailing row
    const int nb = (P - t0 + NB_ - 1) / NB_;
    int bj = 0, rem = blockIdx.x;  // lower triangle of the trailing blocks, by columns: bi >= bj (col_only: the first column)
    if (!col_only)
        while (rem >= nb - bj) { rem -= nb - bj; ++bj; }
    const int bi = bj + rem;
    const int I0 = t0 + bi * NB_, J0 = t0 + bj * NB_;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ci = lane & 15, kr = lane >> 4;
    const bool diag = bi == bj;
    if (threadIdx.x == 0) ready = 0;
    d4_t acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = (d4_t){0., 0., 0., 0.};
    for (int pp = 0; pp < np; ++pp) {
        const int kk = k0 + pp * NB_;
        const double *Linv = LinvAll + (size_t)(kk / NB_) * NB_ * NB_;
        if (pp > 0) __syncthreads();  // (every wave is through with the first panel's L_ik, L_jk)
        for (int i = wv; i < NB_; i += 4) R0[i][lane] = Linv[i * NB_ + lane];
        __syncthreads();
        // rows 16 wv .. 16 wv + 15 of L_ik = A_ik Linv^T and of L_jk: D[r][c] = sum_m A[r][m] Linv[c][m] (m <= c)
        d4_t li[4], lj[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) li[b] = lj[b] = (d4_t){0., 0., 0., 0.};
        {
            const int ra = I0 + 16 * wv + ci, rb = J0 + 16 * wv + ci;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int col = kk + 4 * s + kr;
                const double fa = ra < P ? G[(size_t)ra * P + col] : 0.;
                const double fb = (!diag && rb < P) ? G[(size_t)rb * P + col] : 0.;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (s <= 4 * b + 3) {  // (Linv is lower triangular: columns 16 b .. 16 b + 15 end at m = 16 b + 15)
                        const double fl = R0[16 * b + ci][4 * s + kr];
                        li[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fl, li[b], 0, 0, 0);
                        if (!diag) lj[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb, fl, lj[b], 0, 0, 0);
                    }
            }
        }
        __syncthreads();  // every wave is through with Linv: R0 <- L_ik, R1 <- L_jk
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                R0[16 * wv + kr + 4 * r][16 * b + ci] = li[b][r];
                if (!diag) R1[16 * wv + kr + 4 * r][16 * b + ci] = lj[b][r];
            }
        __syncthreads();
        double (*RJ)[LDP_] = diag ? R0 : R1;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const double fa = R0[16 * wv + ci][4 * s + kr];
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, RJ[16 * b + ci][4 * s + kr], acc[b], 0, 0, 0);
        }
        if (bj == 0) {  // L_ik, transposed, into block (k, i) of the upper triangle (the first of two panels: stored by the column launch)
            if (pp == np - 1)
                for (int m = wv; m < NB_; m += 4)
                    if (I0 + lane < P) G[(size_t)(kk + m) * P + I0 + lane] = R0[lane][m];
        }
    }
    const bool next_diag = diag && bi == 0;  // tile (k+1, k+1): factored here
    double nv[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = I0 + 16 * wv + kr + 4 * r, j = J0 + 16 * b + ci;
            nv[b][r] = 0.;
            if (i < P && j < P && (!diag || j <= i)) {
                nv[b][r] = G[(size_t)i * P + j] - acc[b][r];
                G[(size_t)i * P + j] = nv[b][r];
            }
        }
    if (!next_diag) return;
    __syncthreads();  // (uniform: the whole workgroup is here) R0 and R1 are free
    const int nbk = min(NB_, P - t0);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * wv + kr + 4 * r, j = 16 * b + ci;
            R1[i][j] = (i < nbk && j < nbk) ? nv[b][r] : (i == j ? 1. : 0.);
        }
    __syncthreads();
    bf_chol64_two_waves(R1, R0, &ready, P, t0, nbk, G, LinvAll + (size_t)(t0 / NB_) * NB_ * NB_, info);
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
    hipLaunchKernelGGL(bf_chol_first_kernel, dim3(1), dim3(256), 0, st, P, G, LinvAll, info);
    for (int k0 = 0; k0 + NB_ < P;) {
        const int nb = (P - k0 - NB_ + NB_ - 1) / NB_;   // blocks of rows below panel k0
        if (nb >= 8 && !bf_tune().chol_one_panel) {       // two panels per pass over the trailing matrix (small ones fit the caches)
            hipLaunchKernelGGL(bf_chol_panel_kernel, dim3(nb), dim3(256), 0, st, P, k0, 1, 1, G, LinvAll, info);
            const int nb2 = nb - 1;
            hipLaunchKernelGGL(bf_chol_panel_kernel, dim3(nb2 * (nb2 + 1) / 2), dim3(256), 0, st, P, k0, 2, 0, G, LinvAll, info);
            k0 += 2 * NB_;
        } else {
            hipLaunchKernelGGL(bf_chol_panel_kernel, dim3(nb * (nb + 1) / 2), dim3(256), 0, st, P, k0, 1, 0, G, LinvAll, info);
            k0 += NB_;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Triangular solves: ONE launch per sweep, one workgroup per 64-row block, the blocks of the solution handed from
// workgroup to workgroup as they become final (round 2 had one launch per panel and sweep: 68 dependent launches per
// solve, 2.65 ms of the fit).  Workgroup b of the forward sweep takes L_bk y_k out of its rows for k = 0 .. b-1 in
// order, as soon as block k is there, then forms y_b = L_bb^-1 r_b with the kept inverse and publishes it; the
// backward sweep runs the same way from the last block up, on the transposed factor.  The factor is read-only here;
// what is exchanged -- 64 doubles per block and right-hand side -- goes through an exchange buffer with agent-scope
// atomics (sc1 loads and stores: the eight XCDs' L2 caches are not coherent for plain accesses inside a launch), and
// the data is its own flag: the buffer holds a NaN with a payload no computation produces until the value is written,
// and a reader spins on the value it needs (one trip to memory per hop instead of flag-then-data).  Each sweep clears
// the OTHER sweep's buffer for the launch after it.  A workgroup takes its block from a ticket counter, so it only
// ever waits for workgroups that started before it: no assumption on how many are resident or in which order they
// are dispatched.  Every sum runs in a fixed order: the result does not depend on timing.
// Up to 4 right-hand sides per launch (columns q0 .. q0+mq-1 of r (P, m)), solved in place.
// ---------------------------------------------------------------------------------------------------
#define TRSV_MQ_ 4
#define BF_UNSET_ 0x7FF8DEAD7FF8DEADll  // (both halves equal: set with hipMemsetD32Async)
__device__ inline double bf_ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void bf_st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// sync: [0] ticket counter, [1] finished workgroups.  xch: this sweep's exchange buffer [block][4][64]; xch_next: the
// other sweep's, cleared here
template <bool BWD>
__global__ __launch_bounds__(256) void bf_trsv_flow_kernel(int P, int m, int q0, int mq, const double *__restrict__ G,
                                                         const double *__restrict__ LinvAll, const double *__restrict__ dsc,
                                                         int scale, double *__restrict__ r, int *sync, double *xch,
                                                         double *__restrict__ xch_next) {
    __shared__ double T[NB_][LDP_];
    __shared__ double Li[NB_][LDP_];
    __shared__ double vk[TRSV_MQ_][NB_];
    __shared__ double part[4][TRSV_MQ_][NB_];
    __shared__ int s_ticket;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nb = (P + NB_ - 1) / NB_;
    if (t == 0) s_ticket = atomicAdd(&sync[0], 1);
    __syncthreads();
    const int b = BWD ? nb - 1 - s_ticket : s_ticket;
    const int b0 = b * NB_;
    xch_next[(size_t)b * (TRSV_MQ_ * NB_) + t] = __longlong_as_double(BF_UNSET_);
    // the block's own inverse and right-hand side: on their way while the blocks before it are taken out
    const double *Linv = LinvAll + (size_t)b * NB_ * NB_;
    double lpre[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) lpre[e] = Linv[(wv + 4 * e) * NB_ + lane];
    double own = 0.;
    if (wv < mq && b0 + lane < P) {
        own = r[(size_t)(b0 + lane) * m + q0 + wv];
        if (!BWD && scale) own *= dsc[b0 + lane];  // (forward sweep of a refinement step: the residual comes unscaled)
    }
    // tile (b, k) of the sweep's operator, element (e, lane) for e = wv, wv + 4, ..:
    //   forward  L_bk   = (block (k, b) of the upper triangle)^T:  T[c][row] = G[(k0 + c) P + b0 + row]
    //   backward L_kb^T =  block (b, k) of the upper triangle:     T[row][c] = G[(b0 + row) P + k0 + c]
    double pre[16];
    auto fetch = [&](int k) {
        const int k0 = k * NB_;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int a = wv + 4 * e;
            const int row = BWD ? b0 + a : k0 + a, col = BWD ? k0 + lane : b0 + lane;
            pre[e] = (row < P && col < P) ? G[(size_t)row * P + col] : 0.;
        }
    };
    double accq[TRSV_MQ_];  // thread (lane = row of the block, wave = 16 of a tile's 64 columns)
#pragma unroll
    for (int q = 0; q < TRSV_MQ_; ++q) accq[q] = 0.;
    const int n_k = BWD ? nb - 1 - b : b;
    if (n_k > 0) fetch(BWD ? nb - 1 : 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) Li[wv + 4 * e][lane] = lpre[e];
    for (int kk = 0; kk < n_k; ++kk) {
        const int k = BWD ? nb - 1 - kk : kk;
        __syncthreads();  // the tile and the block before this one have been used
#pragma unroll
        for (int e = 0; e < 16; ++e) T[wv + 4 * e][lane] = pre[e];
        if (kk + 1 < n_k) fetch(BWD ? k - 1 : k + 1);  // (in flight across the wait below)
        if (wv < mq) {
            const double *src = xch + (size_t)k * (TRSV_MQ_ * NB_) + t;
            double v = bf_ld_agent(src);
            while (__double_as_longlong(v) == BF_UNSET_) {
                __builtin_amdgcn_s_sleep(1);
                v = bf_ld_agent(src);
            }
            vk[wv][lane] = v;
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int c = 16 * wv + cc;
            const double tv = BWD ? T[lane][c] : T[c][lane];
#pragma unroll
            for (int q = 0; q < TRSV_MQ_; ++q)
                if (q < mq) accq[q] += tv * vk[q][c];
        }
    }
    // the four waves' partial sums in wave order, then the product with the block's inverse
#pragma unroll
    for (int q = 0; q < TRSV_MQ_; ++q)
        if (q < mq) part[wv][q][lane] = accq[q];
    __syncthreads();
    if (wv < mq) vk[wv][lane] = own - (((part[0][wv][lane] + part[1][wv][lane]) + part[2][wv][lane]) + part[3][wv][lane]);
    __syncthreads();
    if (wv < mq) {
        const int q = wv, c = lane;
        double s = 0.;
        if (BWD) {
            for (int i = c; i < NB_; ++i) s += Li[i][c] * vk[q][i];  // x = L_bb^-T y
        } else {
            for (int i = 0; i <= c; ++i) s += Li[c][i] * vk[q][i];  // y = L_bb^-1 r
        }
        // (rows beyond P: zeros, the padding of the inverse is the identity.  A NaN that reaches here -- from the inputs or
        // from the blocks of a failed factorisation -- is published as the canonical quiet NaN: a payload can never be the
        // "not yet" marker, so a reader's spin always ends)
        bf_st_agent(xch + (size_t)b * (TRSV_MQ_ * NB_) + t, (s != s) ? __longlong_as_double(0x7FF8000000000000ll) : s);
        if (b0 + c < P) r[(size_t)(b0 + c) * m + q0 + q] = s;
    }
    if (t == 0) {
        if (atomicAdd(&sync[1], 1) == nb - 1) {  // the last one out resets the counters for the next launch
            __hip_atomic_store(&sync[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static int ensure_flow(bfhip_ctx *ctx, int nb) {
    if (ctx->flow_cap < nb) {
        BF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->flow) BF_HIP_CHECK(hipFree(ctx->flow));
        ctx->flow = NULL;
        ctx->flow_cap = 0;
        const int cap = nb < 256 ? 256 : 2 * nb;
        // [2 ints of counters, padded to 64 bytes][forward exchange buffer][backward exchange buffer]
        const size_t n_x = (size_t)cap * TRSV_MQ_ * NB_;
        BF_HIP_CHECK(hipMalloc((void **)&ctx->flow, 64 + 2 * n_x * sizeof(double)));
        BF_HIP_CHECK(hipMemsetAsync(ctx->flow, 0, 64, ctx->stream));
        BF_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)((char *)ctx->flow + 64), 0x7FF8DEAD, 4 * n_x, ctx->stream));
        ctx->flow_cap = cap;
    }
    return 0;
}

// x = (D L L^T D)^-1 r with the kept factor (the transposed blocks in the upper triangle of G), the inverses of its
// diagonal blocks and the scales D = diag(dsc): r is overwritten by x.  scale_in: r comes unscaled (a refinement
// step's A^T S); otherwise the caller has scaled it already (bf_scale_kernel)
static int chol_apply(bfhip_ctx *ctx, int P, int m, const double *L, const double *dsc, const double *LinvAll, double *r,
                      bool scale_in) {
    hipStream_t st = ctx->stream;
    const int nb = (P + NB_ - 1) / NB_;
    if (int rc = ensure_flow(ctx, nb)) return rc;
    int *sync = (int *)ctx->flow;
    double *xf = (double *)((char *)ctx->flow + 64), *xb = xf + (size_t)ctx->flow_cap * TRSV_MQ_ * NB_;
    for (int q0 = 0; q0 < m; q0 += TRSV_MQ_) {
        const int mq = m - q0 < TRSV_MQ_ ? m - q0 : TRSV_MQ_;
        hipLaunchKernelGGL(bf_trsv_flow_kernel<false>, dim3(nb), dim3(256), 0, st, P, m, q0, mq, L, LinvAll, dsc, scale_in ? 1 : 0, r,
                           sync, xf, xb);
        hipLaunchKernelGGL(bf_trsv_flow_kernel<true>, dim3(nb), dim3(256), 0, st, P, m, q0, mq, L, LinvAll, dsc, 0, r, sync, xb, xf);
    }
    hipLaunchKernelGGL(bf_unscale_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, m, r, dsc);
    return 0;
}

extern "C" int bfhip_solve_spd(bfhip_ctx *ctx, int P, int m, double *G, double *r, int *info) {
    BfDeviceGuard dev_guard(ctx);
    if (!ctx || P < 1 || m < 1 || !G || !r || !info) return bf_set_error(BFHIP_ERR_ARG, "bfhip_solve_spd: invalid argument");
    const size_t n_inv = (size_t)((P + NB_ - 1) / NB_) * NB_ * NB_;
    if (int rc = ensure_scratch(ctx, ((size_t)P + n_inv) * sizeof(double))) return rc;
    double *dsc = (double *)ctx->scratch, *LinvAll = dsc + P;
    if (int rc = solve_spd_impl(ctx, P, m, G, r, info, dsc, LinvAll)) return rc;
    if (int rc = chol_apply(ctx, P, m, G, dsc, LinvAll, r, false)) return rc;
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
    if (int rc = ensure_scratch(ctx, (n_atb + P + n_inv) * sizeof(double))) return rc;
    double *part = (double *)ctx->scratch, *dsc = part + n_atb, *LinvAll = dsc + P;
    if (int rc = solve_spd_impl(ctx, P, m, G, c, info, dsc, LinvAll)) return rc;
    if (int rc = chol_apply(ctx, P, m, G, dsc, LinvAll, c, false)) return rc;
    double *S = work, *dc = work + (size_t)n * m;
    hipStream_t st = ctx->stream;
    for (int it = 0; it < n_refine; ++it) {
        hipLaunchKernelGGL(bf_resid_kernel, dim3((n + 3) / 4), dim3(256), 0, st, n, P, m, A, lda, B, c, S);
        hipLaunchKernelGGL(bf_atb_kernel, dim3((P + 127) / 128, m, ATB_SEG_), dim3(128), 0, st, n, P, m, A, lda, S, part);
        hipLaunchKernelGGL(bf_atb_reduce_kernel, dim3((P + 127) / 128, m), dim3(128), 0, st, P, m, part, dc);
        if (int rc = chol_apply(ctx, P, m, G, dsc, LinvAll, dc, true)) return rc;
        hipLaunchKernelGGL(bf_axpy_kernel, dim3((P * m + 255) / 256), dim3(256), 0, st, P * m, c, dc);
    }
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}
