// Microprobe: v_mfma_f64_4x4x4_4b_f64 lane maps and issue rate (diagnostic tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void layout_k(const double* A, const double* B, double* D) {
  int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
template <int NACC>
__global__ void rate(double* out, int iters) {
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9;
  double c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int i = 0; i < NACC; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}
int main() {
  // layout: feed A[l] = unique powers, find which (a-lane, b-lane) pairs contribute to each output lane
  std::vector<double> A(64), B(64), D(64);
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 512));
  // for each output lane, determine contributing A lanes: set A = e_i (one-hot), B = all ones
  printf("D lane <- sum over listed A lanes (B = 1):\n");
  std::vector<std::vector<int>> fromA(64), fromB(64);
  for (int i = 0; i < 64; ++i) {
    for (int k = 0; k < 64; ++k) { A[k] = (k == i); B[k] = 1.0; }
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout_k<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; ++l) if (D[l] != 0) fromA[l].push_back(i);
    for (int k = 0; k < 64; ++k) { B[k] = (k == i); A[k] = 1.0; }
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout_k<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; ++l) if (D[l] != 0) fromB[l].push_back(i);
  }
  for (int l = 0; l < 64; l += 1) { if (l < 20 || l % 16 == 0) { printf("  D[%2d]: A lanes", l); for (int v : fromA[l]) printf(" %d", v); printf(" | B lanes"); for (int v : fromB[l]) printf(" %d", v); printf("\n"); } }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount; int iters = 20000;
  double* dout; CK(hipMalloc(&dout, ((size_t)cus * 1024 + 8) * 8));
  for (int wps = 1; wps <= 4; wps *= 2) {
    int threads = 256 * wps;
    for (int nacc : {1, 4}) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      auto launch = [&] { if (nacc == 1) rate<1><<<cus, threads>>>(dout, iters); else rate<4><<<cus, threads>>>(dout, iters); };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double cyc; CK(hipMemcpy(&cyc, dout + (size_t)cus * threads, 8, hipMemcpyDeviceToHost));
      double flop = (double)cus * 4 * wps * iters * nacc * 512.0;
      printf("4x4x4_4b: %d wave/SIMD, %d acc: %.3f ms, %.2f TFLOP/s, wave0 ticks/instr %.2f\n", wps, nacc, ms, flop / ms * 1e-9, cyc / (iters * nacc));
    }
  }
  return 0;
}
