// Microprobe (round 5): the latencies a lone chain's leapfrog step is made of on gfx950 -- dependent FP64 VALU / 4x4x4 MFMA
// chains, the wave reductions, LDS round trips, workgroup barriers by wave count, and a flag hand-off between two waves
// through LDS.  One workgroup on one CU; ticks are s_memtime (core clock).
// build: hipcc -O3 --offload-arch=gfx950 -o latency_probe latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline double dpp_ror8(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x128, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x128, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double dpp_ror4(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x124, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x124, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double rfl(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ inline double wsum(double v) {
    v = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1., 0., 0, 0, 0);
    v = __builtin_amdgcn_mfma_f64_4x4x4f64(1., v, 0., 0, 0, 0);
    v += dpp_ror8(v);
    v += dpp_ror4(v);
    return rfl(v);
}

// mode: which dependent chain
template <int MODE>
__global__ void chain_k(double *out, int iters, double seed) {
    __shared__ double sh[1024];
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-9 + seed, b = 1.0 - l * 1e-9, c = seed;
    sh[threadIdx.x] = a;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) c = __builtin_fma(a, c, b);
        if (MODE == 1) c = c + a;
        if (MODE == 2) c = c * a;
        if (MODE == 3) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        if (MODE == 4) { c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); c = c + a; }  // mfma -> valu -> mfma
        if (MODE == 5) c = wsum(c * b);
        if (MODE == 6) { sh[threadIdx.x] = c; __builtin_amdgcn_s_waitcnt(0xc07f); c = sh[threadIdx.x ^ 1] + 1.; }  // LDS write -> read (other lane)
        if (MODE == 7) c = exp(c * 1e-3);
        if (MODE == 8) c = log(c + 2.);
        if (MODE == 9) c = sqrt(c + 2.);
        if (MODE == 10) c = b / (c + 2.);
        if (MODE == 11) c = c + dpp_ror8(c);
        if (MODE == 12) c = rfl(c) + a;
        if (MODE == 13) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, (__attribute__((ext_vector_type(4))) double){c, c, c, c}, 0, 0, 0)[0];
        if (MODE == 14) { float f = (float)c; f = __builtin_fmaf(f, 1.0001f, 0.5f); c = (double)f; }
        if (MODE == 15) { c = sh[(threadIdx.x + (int)c) & 1023]; }   // dependent LDS read
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

// barriers per wave count
__global__ void barrier_k(double *out, int iters) {
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) __syncthreads();
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0);
}
// bare s_barrier (no waitcnt / fence)
__global__ void sbarrier_k(double *out, int iters) {
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0);
}
// LDS write, barrier, LDS read of another wave's value (the exchange pattern of a trip)
__global__ void xchg_k(double *out, int iters) {
    __shared__ double sh[1024];
    const int nt = blockDim.x;
    double c = threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        sh[threadIdx.x] = c;
        __syncthreads();
        c = sh[(threadIdx.x + 64) % nt] + 1.;
        __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[1 + threadIdx.x] = c;
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0);
}

// flag ping-pong between wave 0 and wave WB through LDS (no barrier): round trips
__global__ void pingpong_k(double *out, int iters, int wb, int sleep) {
    __shared__ volatile int flag[2];
    __shared__ volatile double pay[2][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (threadIdx.x < 2) flag[threadIdx.x] = 0;
    __syncthreads();
    double c = l;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (w == 0) {
        for (int it = 1; it <= iters; ++it) {
            pay[0][l] = c;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (l == 0) flag[0] = it;
            while (flag[1] != it) { if (sleep) __builtin_amdgcn_s_sleep(1); }
            c = pay[1][l] + 1.;
        }
    } else if (w == wb) {
        for (int it = 1; it <= iters; ++it) {
            while (flag[0] != it) { if (sleep) __builtin_amdgcn_s_sleep(1); }
            c = pay[0][l] + 1.;
            pay[1][l] = c;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (l == 0) flag[1] = it;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[1 + threadIdx.x] = c;
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0);
}

// a wave's dependent FP64 VALU chain while the other waves of the workgroup run 4x4x4 MFMA chains (shared pipe?)
__global__ void contend_k(double *out, int iters, int mode) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9, c = 0., c2 = 0.;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (w == 0) {
        for (int it = 0; it < iters; ++it) c = __builtin_fma(a, c, b);
    } else {
        for (int it = 0; it < iters; ++it) {
            if (mode == 1) { c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); }
            if (mode == 2) { c = __builtin_fma(a, c, b); c2 = __builtin_fma(a, c2, b); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[1 + threadIdx.x] = c + c2;
    if (l == 0) out[2048 + w] = (double)(t1 - t0);
}

int main() {
    double *dout;
    CK(hipMalloc(&dout, 8192 * 8));
    const int iters = 4000;
    const char *names[] = {"v_fma_f64 dependent", "v_add_f64 dependent", "v_mul_f64 dependent", "mfma_4x4x4 dependent", "mfma_4x4x4 -> add -> mfma",
                           "wave_sum (2 mfma + 2 dpp-add + rfl) of a product", "LDS write -> wait -> read other lane -> add", "exp (inline libm)", "log", "sqrt", "division",
                           "dpp row_ror + add", "readfirstlane pair + add", "mfma_16x16x4 dependent", "f64->f32 fma ->f64", "dependent LDS read"};
    for (int waves : {1, 4}) {
        printf("-- %d wave(s) in the workgroup (one CU)\n", waves);
        for (int m = 0; m < 16; ++m) {
            double cyc = 0;
#define RUN(M) case M: chain_k<M><<<1, 64 * waves>>>(dout, iters, 0.); break;
            for (int rep = 0; rep < 2; ++rep) {
                switch (m) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) }
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(&cyc, dout + 64 * waves, 8, hipMemcpyDeviceToHost));
            printf("%-52s %8.1f ticks / iteration\n", names[m], cyc / iters);
        }
    }
    for (int waves : {1, 2, 4, 8, 16}) {
        double c1, c2, c3;
        for (int rep = 0; rep < 2; ++rep) { barrier_k<<<1, 64 * waves>>>(dout, iters); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(&c1, dout, 8, hipMemcpyDeviceToHost));
        for (int rep = 0; rep < 2; ++rep) { sbarrier_k<<<1, 64 * waves>>>(dout, iters); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(&c2, dout, 8, hipMemcpyDeviceToHost));
        for (int rep = 0; rep < 2; ++rep) { xchg_k<<<1, 64 * waves>>>(dout, iters); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(&c3, dout, 8, hipMemcpyDeviceToHost));
        printf("%2d waves: __syncthreads %.1f  bare s_barrier %.1f  write+sync+read+sync %.1f ticks\n", waves, c1 / iters, c2 / iters, c3 / iters);
    }
    for (int wb : {1, 4}) for (int sleep : {0, 1}) {
        double c1;
        for (int rep = 0; rep < 2; ++rep) { pingpong_k<<<1, 64 * 8>>>(dout, iters, wb, sleep); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(&c1, dout, 8, hipMemcpyDeviceToHost));
        printf("flag ping-pong wave 0 <-> wave %d (8 waves resident, sleep %d): %.1f ticks per round trip (two hand-offs with a 64-double payload)\n", wb, sleep, c1 / iters);
    }
    for (int waves : {1, 5, 9}) for (int mode : {1, 2}) {
        std::vector<double> t(16);
        for (int rep = 0; rep < 2; ++rep) { contend_k<<<1, 64 * waves>>>(dout, iters, mode); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(t.data(), dout + 2048, 16 * 8, hipMemcpyDeviceToHost));
        printf("contention, %d waves, others run %s: wave 0 fma chain %.1f ticks/op, wave 1 %.1f ticks/iteration (2 ops)\n", waves, mode == 1 ? "2 mfma_4x4x4 chains" : "2 fma chains",
               t[0] / iters, waves > 1 ? t[1] / iters : 0.);
    }
    return 0;
}
