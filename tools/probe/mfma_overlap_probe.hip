// Probe: can VALU work overlap with v_mfma_f64_16x16x4_f64 on gfx950?
//   WPS waves per SIMD run, per iteration, one MFMA (dependent chain through the accumulator) followed by NV
//   dependent f64 FMAs on independent registers.  If the FMAs run in the MFMA's shadow the time per iteration
//   stays at the MFMA's until NV * (FMA latency) exceeds it; if not, the two add up.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap_probe mfma_overlap_probe.hip && ./mfma_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV, int NMF>
__global__ void probe(double *out, int iters) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane, v = 1e-3 * lane;
  d4 c = {0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NMF; ++k) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    asm volatile("" : "+v"(v));
#pragma unroll
    for (int k = 0; k < NV; ++k) v = __builtin_fma(v, 1.0000001, 1e-9);
    asm volatile("" : "+v"(v));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v + c[0] + c[1] + c[2] + c[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}

template <int NV, int NMF>
int run(double *dout, int wps, int iters) {
  // one workgroup per CU, wps waves per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<NV, NMF><<<256, 256 * wps>>>(dout, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  probe<NV, NMF><<<256, 256 * wps>>>(dout, iters);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double ticks; CK(hipMemcpy(&ticks, dout + (size_t)256 * 256 * wps, 8, hipMemcpyDeviceToHost));
  printf("waves/SIMD %d  MFMAs %d  FMAs %3d : %8.1f ticks/iteration  %8.1f ns/iteration\n", wps, NMF, NV, ticks / iters, ms * 1e6 / iters);
  return 0;
}

int main() {
  double *dout; CK(hipMalloc(&dout, (256 * 1024 + 8) * sizeof(double)));
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    run<0, 1>(dout, wps, iters); run<8, 0>(dout, wps, iters); run<8, 1>(dout, wps, iters); run<16, 1>(dout, wps, iters);
    run<32, 0>(dout, wps, iters); run<32, 1>(dout, wps, iters); run<64, 1>(dout, wps, iters); run<64, 0>(dout, wps, iters);
  }
  return 0;
}
