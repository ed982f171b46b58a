// Probe: how does v_mfma_f64_16x16x4_f64 round?  Runs one MFMA on random operands and prints the operands and
// results of a few (row, col) entries as hex so that candidate summation orders can be checked exactly on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, const double *C, double *D) {
  int l = threadIdx.x;
  d4 c = {C[l * 4 + 0], C[l * 4 + 1], C[l * 4 + 2], C[l * 4 + 3]};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
  double hA[64], hB[64], hC[256], hD[256];
  srand(7);
  auto rnd = [] { return (rand() / (double)RAND_MAX - 0.5) * exp2((rand() % 40) - 20); };
  for (int i = 0; i < 64; ++i) { hA[i] = rnd(); hB[i] = rnd(); }
  for (int i = 0; i < 256; ++i) hC[i] = rnd();
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 2048, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dC, dD);
  hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
  // entry (row, col): lane l = col + 16 * (row & 3), reg r = row >> 2 holds D[row][col]; A[i][k] at lane i + 16 k; B[k][j] at lane j + 16 k
  int n_seq = 0, n_rev = 0, n_tot = 0;
  for (int row = 0; row < 16; ++row)
    for (int col = 0; col < 16; ++col) {
      int l = col + 16 * (row & 3), r = row >> 2;
      double c = hC[l * 4 + r], d = hD[l * 4 + r];
      double s = c, t = c;
      for (int kk = 0; kk < 4; ++kk) s = fma(hA[row + 16 * kk], hB[col + 16 * kk], s);
      for (int kk = 3; kk >= 0; --kk) t = fma(hA[row + 16 * kk], hB[col + 16 * kk], t);
      n_tot++; n_seq += (s == d); n_rev += (t == d);
      if (row < 2 && col < 3) {
        printf("entry %d %d: c=%a d=%a seq=%a rev=%a\n", row, col, c, d, s, t);
        for (int kk = 0; kk < 4; ++kk) printf("   a=%a b=%a\n", hA[row + 16 * kk], hB[col + 16 * kk]);
      }
    }
  printf("matches: sequential fma k=0..3: %d / %d ; reverse k=3..0: %d / %d\n", n_seq, n_tot, n_rev, n_tot);
  return 0;
}
