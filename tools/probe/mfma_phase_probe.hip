// Probe: the sampler's matvec phase in isolation.  16 waves per CU (4 per SIMD); every loop trip each wave runs
// NM dependent v_mfma_f64_16x16x4_f64 between two workgroup barriers, optionally with the operand loads
// from LDS and the result store to LDS that the sampler does.  Prints cycles per trip (wave 0, s_memtime)
// and wall time per trip.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_phase_probe mfma_phase_probe.hip && ./mfma_phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int MODE, int IDLE = 0>  // IDLE: dependent f64 FMAs run by wave 0 alone between the phases (duty cycle)
// MODE 0: registers only; 1: B operands from LDS; 2: + results to LDS and read back by wave 0
__global__ __launch_bounds__(1024) void phase(double *out, int trips) {
  __shared__ double XB[16 * 65 * 2];
  __shared__ double GB[4 * 16 * 65];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 16 * 65 * 2; i += 1024) XB[i] = 1e-3 * i;
  double a[8];
  for (int s = 0; s < 8; ++s) a[s] = 1.0 + 1e-9 * (lane + s);
  double keep = 0.;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < trips; ++it) {
    if (IDLE > 0 && w == 0) {
#pragma unroll 8
      for (int k = 0; k < IDLE; ++k) keep = __builtin_fma(keep, 1.0000001, 1e-9);
    }
    if (MODE >= 1 && w == 0) XB[(it & 15) * 65 + lane] = keep * 1e-30 + it;  // the chain wave posts its x
    __syncthreads();
    double b[8];
    for (int s = 0; s < 8; ++s) b[s] = MODE >= 1 ? XB[((w & 1) * 16 + s) * 65 + lane] : 1.0 - 1e-9 * (lane + s + it);
    d4 c = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NM; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s & 7], b[s & 7], c, 0, 0, 0);
    if (MODE >= 2) {
      for (int r = 0; r < 4; ++r) GB[((w >> 2) * 16 + (lane & 15)) * 65 + 16 * (w & 3) + 4 * r + (lane >> 4)] = c[r];
    } else {
      keep += c[0] + c[1] + c[2] + c[3];
    }
    __syncthreads();
    if (MODE >= 2 && w == 0) keep += GB[lane] + GB[16 * 65 + lane] + GB[2 * 16 * 65 + lane] + GB[3 * 16 * 65 + lane];
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 1024 + tid] = keep;
  if (tid == 0 && blockIdx.x == 0) out[gridDim.x * 1024] = (double)(t1 - t0);
}

template <int NM, int MODE, int IDLE = 0>
int run(double *dout, int cus, int trips) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  phase<NM, MODE, IDLE><<<cus, 1024>>>(dout, trips);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  phase<NM, MODE, IDLE><<<cus, 1024>>>(dout, trips);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double ticks; CK(hipMemcpy(&ticks, dout + (size_t)cus * 1024, 8, hipMemcpyDeviceToHost));
  printf("NM %2d mode %d idle %3d: %8.1f ns/trip  %8.1f ticks/trip  (%.2f ticks/ns)   per MFMA-slot on a SIMD: %.1f ticks\n", NM, MODE, IDLE,
         ms * 1e6 / trips, ticks / trips, ticks / (ms * 1e6), NM ? ticks / trips / (4. * NM) : 0.);
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount, trips = 20000;
  double *dout; CK(hipMalloc(&dout, ((size_t)cus * 1024 + 8) * 8));
  printf("%d CUs, 16 waves per CU, %d trips\n", cus, trips);
  run<0, 0>(dout, cus, trips); run<1, 0>(dout, cus, trips); run<4, 0>(dout, cus, trips); run<8, 0>(dout, cus, trips); run<16, 0>(dout, cus, trips);
  run<8, 1>(dout, cus, trips); run<8, 2>(dout, cus, trips); run<0, 2>(dout, cus, trips); run<16, 2>(dout, cus, trips);
  run<0, 2, 100>(dout, cus, trips); run<8, 2, 100>(dout, cus, trips); run<16, 2, 100>(dout, cus, trips);
  run<0, 2, 300>(dout, cus, trips); run<8, 2, 300>(dout, cus, trips); run<16, 2, 300>(dout, cus, trips);
  return 0;
}
