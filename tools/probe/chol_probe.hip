// Timing probe of the 64 x 64 diagonal step of the blocked Cholesky (bfhip_fit.hip: bf_chol64_two_waves): variants of the
// broadcast of column j (LDS reads vs v_readlane) and of the depth of the pinned groups.  hipcc -DVAR=n; prints us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define NB_ 64
#define LDP_ 66
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ inline double bf_readlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#ifndef VAR
#define VAR 0
#endif
#ifndef GRP
#define GRP 16
#endif
__global__ __launch_bounds__(256) void k0(int P, double *__restrict__ G, double *__restrict__ Lout, double *__restrict__ Linv, int *__restrict__ info) {
    __shared__ __attribute__((aligned(16))) double S[NB_][LDP_];
    __shared__ __attribute__((aligned(16))) double Lc[NB_][LDP_];
    __shared__ int ready;
    const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < NB_; i += 4) S[i][t] = G[(size_t)i * P + t];
    if (threadIdx.x == 0) ready = 0;
    __syncthreads();
    if (wv == 0) {
        double a[NB_];
#pragma unroll
        for (int c = 0; c < NB_; ++c) a[c] = S[t][c];
        int bad = 0;
#pragma unroll
        for (int j = 0; j < NB_; ++j) {
            double djj = bf_readlane(a[j], j);
            const bool ok = djj > 1e-11;
            bad = (!ok && bad == 0) ? j + 1 : bad;
            djj = ok ? djj : 1.;
            const double rl = rsqrt(djj), ljj = djj * rl;
            int tt = t;
            asm volatile("" : "+v"(tt));
            const double lj = tt == j ? ljj : (tt > j ? a[j] * rl : 0.);
            Lc[j][t] = lj;
            S[t][j] = lj;
            if (t == 0) Lc[j][NB_] = rl;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (t == 0) __hip_atomic_store(&ready, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            if (j + 1 < NB_) {
                a[j + 1] -= lj * bf_readlane(lj, j + 1);
                asm volatile("" : "+v"(a[j + 1]));
            }
#if VAR == 0   // LDS broadcasts, two columns per read
            if ((j & 1) && j + 2 < NB_) a[j + 2] -= lj * Lc[j][j + 2];
#pragma unroll
            for (int c = (j + 3) & ~1; c < NB_; c += 2) {
                const d2_t l2 = *(const d2_t *)&Lc[j][c];
                a[c] -= lj * l2.x;
                a[c + 1] -= lj * l2.y;
                if ((c & (GRP - 1)) == GRP - 2) {
#pragma unroll
                    for (int cc = c & ~(GRP - 1); cc <= c; cc += 2)
                        if (cc > j + 1) asm volatile("" : "+v"(a[cc]), "+v"(a[cc + 1]));
                }
            }
#else          // v_readlane broadcasts
#pragma unroll
            for (int c = j + 2; c < NB_; ++c) {
                a[c] -= lj * bf_readlane(lj, c);
                if ((c & (GRP - 1)) == GRP - 1) {
#pragma unroll
                    for (int cc = c & ~(GRP - 1); cc <= c; ++cc)
                        if (cc > j + 1) asm volatile("" : "+v"(a[cc]));
                }
            }
#endif
        }
        if (bad && t == 0) atomicCAS(info, 0, bad);
        for (int i = 0; i < NB_; ++i) Lout[(size_t)i * P + t] = S[i][t];
    }
#ifndef NO_W1
    else if (wv == 1) {
        double x[NB_];
#pragma unroll
        for (int j = 0; j < NB_; ++j) {
            while (__hip_atomic_load(&ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= j) __builtin_amdgcn_s_sleep(1);
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
#pragma unroll
        for (int j = 0; j < NB_; ++j) Linv[j * NB_ + t] = x[j];
    }
#endif
}

typedef double d4_t __attribute__((ext_vector_type(4)));
// VAR 2: the columns of the block over the four waves (16 each), the inverse by 32 x 32 blocks
__global__ __launch_bounds__(256) void k4(int P, double *__restrict__ G, double *__restrict__ Lout, double *__restrict__ Linv, int *__restrict__ info) {
    __shared__ __attribute__((aligned(16))) double S[NB_][LDP_];
    __shared__ __attribute__((aligned(16))) double Lc[NB_][LDP_];
    __shared__ int ready;
    const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < NB_; i += 4) S[i][t] = G[(size_t)i * P + t];
    if (threadIdx.x == 0) ready = 0;
    __syncthreads();
    const int c0 = 16 * wv;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = S[t][c0 + c];
    __syncthreads();   // (S is overwritten by the factor below)
    int bad = 0;
#pragma unroll
    for (int j = 0; j < NB_; ++j) {
        const int owner = j >> 4, jj = j & 15;
        if (wv == owner) {
            double djj = bf_readlane(a[jj], j);
            const bool ok = djj > 1e-11;
            bad = (!ok && bad == 0) ? j + 1 : bad;
            djj = ok ? djj : 1.;
#ifdef RSQ_FAST
            double rl = __builtin_amdgcn_rsq(djj);
            { const double e = __builtin_fma(-djj * rl, rl, 1.); rl = __builtin_fma(rl * e, __builtin_fma(e, 0.375, 0.5), rl); }
            const double ljj = djj * rl;
#else
            const double rl = rsqrt(djj), ljj = djj * rl;
#endif
            int tt = t;
            asm volatile("" : "+v"(tt));
            const double lj = tt == j ? ljj : (tt > j ? a[jj] * rl : 0.);
#ifdef FENCE_DEFER
            // the flag of the step before (its LDS writes were issued a whole step ago), then this step's writes
            if (jj > 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (t == 0) __hip_atomic_store(&ready, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            Lc[j][t] = lj;
            S[t][j] = lj;
            if (t == 0) Lc[j][NB_] = rl;
            if (jj == 15) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (t == 0) __hip_atomic_store(&ready, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#else
            Lc[j][t] = lj;
            S[t][j] = lj;
            if (t == 0) Lc[j][NB_] = rl;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (t == 0) __hip_atomic_store(&ready, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
#pragma unroll
            for (int cc = jj + 1; cc < 16; ++cc) {
                a[cc] -= lj * bf_readlane(lj, c0 + cc);
                asm volatile("" : "+v"(a[cc]));
            }
        } else if (wv > owner) {
            while (__hip_atomic_load(&ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= j) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const double lj = Lc[j][t];
#pragma unroll
            for (int cc = 0; cc < 16; cc += 2) {
                const d2_t l2 = *(const d2_t *)&Lc[j][c0 + cc];
                a[cc] -= lj * l2.x;
                a[cc + 1] -= lj * l2.y;
            }
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) asm volatile("" : "+v"(a[cc]));
        }
    }
    if (wv == 3 && bad && t == 0) atomicCAS(info, 0, bad);   // (only the owner of a step sees its pivot; kept simple in the probe)
    // diagonal blocks of the inverse: X11 by wave 0 (rows 0..31), X22 by wave 1 (rows 32..63), lane c < 32 = column
    double x[32];
#ifdef NO_INV
    for (int s = 0; s < 32; ++s) x[s] = 0.;
    if (false) {
#else
    if (wv < 2) {
#endif
        const int r0 = 32 * wv;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int j = r0 + s;
            while (__hip_atomic_load(&ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= j) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int tt = t;
            asm volatile("" : "+v"(tt));
            double sm[4] = {tt == s ? 1. : 0., 0., 0., 0.};
#pragma unroll
            for (int k = 0; k + 1 < s; k += 2) {
                const d2_t l2 = *(const d2_t *)&S[j][r0 + k];
                sm[(k >> 1) & 1] -= l2.x * x[k];
                sm[2 + ((k >> 1) & 1)] -= l2.y * x[k + 1];
            }
            if (s & 1) sm[0] -= S[j][r0 + s - 1] * x[s - 1];
            x[s] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) * Lc[j][NB_];
            asm volatile("" : "+v"(x[s]));
        }
    }
    __syncthreads();   // the factor is complete (S row-major); Lc is free
    if (wv == 3) for (int i = 0; i < NB_; ++i) Lout[(size_t)i * P + t] = S[i][t];
    if (wv < 2) {
        const int r0 = 32 * wv;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (t < 32) Lc[r0 + s][r0 + t] = x[s];
            const double v = t < 32 ? x[s] : 0.;
            if (wv == 0) Linv[(r0 + s) * NB_ + t] = v;                       // rows 0..31: X11 | 0
            else if (t < 32) Linv[(r0 + s) * NB_ + r0 + t] = x[s];           // rows 32..63, columns 32..63: X22
        }
    }
    __syncthreads();
    // X21 = -X22 (L21 X11): one 16 x 16 tile per wave
    const int ci = t & 15, kr = t >> 4, ta = wv >> 1, tb = wv & 1;
    d4_t acc = {0., 0., 0., 0.};
#pragma unroll
    for (int s = 0; s < 8; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[32 + 16 * ta + ci][4 * s + kr], Lc[4 * s + kr][16 * tb + ci], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Lc[32 + 16 * ta + kr + 4 * r][16 * tb + ci] = acc[r];
    __syncthreads();
    d4_t acc2 = {0., 0., 0., 0.};
#pragma unroll
    for (int s = 0; s < 8; ++s)
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lc[32 + 16 * ta + ci][32 + 4 * s + kr], Lc[32 + 4 * s + kr][16 * tb + ci], acc2, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Linv[(32 + 16 * ta + kr + 4 * r) * NB_ + 16 * tb + ci] = -acc2[r];
}
__global__ void kempty() {}

#if VAR == 2
#define KERN k4
#else
#define KERN k0
#endif
int main() {
    const int P = 64;
    std::vector<double> A(P * P), B(P * P, 0.);
    unsigned s = 12345;
    for (auto &v : A) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536. - 0.5; }
    for (int i = 0; i < P; ++i) for (int j = 0; j < P; ++j) { double acc = i == j ? 1. : 0.; for (int k = 0; k < P; ++k) acc += A[i * P + k] * A[j * P + k] / P; B[i * P + j] = acc; }
    double *dG, *dL, *dI; int *dinfo;
    hipMalloc(&dG, P * P * 8); hipMalloc(&dL, P * P * 8); hipMalloc(&dI, P * P * 8); hipMalloc(&dinfo, 4);
    hipMemcpy(dG, B.data(), P * P * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms_e, ms_k;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kempty, dim3(1), dim3(256), 0, 0);
    hipEventRecord(e0); for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kempty, dim3(1), dim3(256), 0, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_e, e0, e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(KERN, dim3(1), dim3(256), 0, 0, P, dG, dL, dI, dinfo);
    hipEventRecord(e0); for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(KERN, dim3(1), dim3(256), 0, 0, P, dG, dL, dI, dinfo); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_k, e0, e1);
    std::vector<double> L(P * P), I(P * P);
    hipMemcpy(L.data(), dL, P * P * 8, hipMemcpyDeviceToHost); hipMemcpy(I.data(), dI, P * P * 8, hipMemcpyDeviceToHost);
    double e_l = 0., e_i = 0.;
    for (int i = 0; i < P; ++i) for (int j = 0; j < P; ++j) {
        double acc = 0., acc2 = 0.;
        for (int k = 0; k < P; ++k) { acc += L[i * P + k] * L[j * P + k]; acc2 += I[i * P + k] * L[k * P + j]; }
        e_l = fmax(e_l, fabs(acc - B[i * P + j])); e_i = fmax(e_i, fabs(acc2 - (i == j)));
    }
    printf("VAR %d GRP %d: %.2f us per launch (empty launch %.2f us)  |L L^T - A| %.1e  |Linv L - I| %.1e\n", VAR, GRP, ms_k * 5., ms_e * 5., e_l, e_i);
    return 0;
}
