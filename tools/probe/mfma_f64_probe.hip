// Microprobe for gfx950: operand/result lane maps and issue rate of
// v_mfma_f64_16x16x4_f64, and the rate of v_fma_f64.  Diagnostic tool, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void layout_k(const double* A, const double* B, double* D) {
  // A is 16x4 row-major, B is 4x16 row-major, D 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ void rate_mfma(double* out, int iters) {
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9;
  d4 c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = (d4){0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

template <int NACC>
__global__ void rate_fma(double* out, int iters) {
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-9, b = 1e-9 * l;
  double c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_fma(c[i], a, b);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

template <typename F>
static void timeit(const char* name, F launch, double flop_per_launch, double* dout, size_t nout, int n_instr_per_wave) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double cyc; CK(hipMemcpy(&cyc, dout + nout, 8, hipMemcpyDeviceToHost));
  printf("%-40s %8.3f ms  %8.2f TFLOP/s  wave0 cycles(memtime ticks)/instr = %.2f\n", name, ms, flop_per_launch / ms * 1e-9, cyc / n_instr_per_wave);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  clock %d kHz  gcn %s\n", p.name, p.multiProcessorCount, p.clockRate, p.gcnArchName);
  // layout
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 131;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + k * 17 + j * 1009;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  layout_k<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
  printf("layout check: %d mismatches of 256 (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)+4r][col=l&15])\n", bad);
  // rates
  int cus = p.multiProcessorCount;
  size_t nout = (size_t)cus * 8 * 256;
  double* dout; CK(hipMalloc(&dout, (nout + 8) * 8));
  int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    int threads = 256 * wps;  // wps waves per SIMD, one block per CU
    double nwaves = (double)cus * 4 * wps;
    printf("-- %d wave(s) per SIMD, %d blocks of %d threads\n", wps, cus, threads);
    timeit("mfma_f64_16x16x4 1 acc (dependent)", [&] { rate_mfma<1><<<cus, threads>>>(dout, iters); }, nwaves * iters * 1 * 2048.0, dout, (size_t)cus * threads, iters * 1);
    timeit("mfma_f64_16x16x4 4 acc", [&] { rate_mfma<4><<<cus, threads>>>(dout, iters); }, nwaves * iters * 4 * 2048.0, dout, (size_t)cus * threads, iters * 4);
    timeit("mfma_f64_16x16x4 8 acc", [&] { rate_mfma<8><<<cus, threads>>>(dout, iters); }, nwaves * iters * 8 * 2048.0, dout, (size_t)cus * threads, iters * 8);
    timeit("v_fma_f64 1 acc (dependent)", [&] { rate_fma<1><<<cus, threads>>>(dout, iters); }, nwaves * iters * 1 * 128.0, dout, (size_t)cus * threads, iters * 1);
    timeit("v_fma_f64 8 acc", [&] { rate_fma<8><<<cus, threads>>>(dout, iters); }, nwaves * iters * 8 * 128.0, dout, (size_t)cus * threads, iters * 8);
    timeit("v_fma_f64 16 acc", [&] { rate_fma<16><<<cus, threads>>>(dout, iters); }, nwaves * iters * 16 * 128.0, dout, (size_t)cus * threads, iters * 16);
  }
  return 0;
}
