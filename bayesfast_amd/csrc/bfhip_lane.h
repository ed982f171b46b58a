// bfhip_lane.h -- the handful of wave / workgroup primitives the group sampler kernel (bfhip_group.h) is written in.
//
// Device build (hipcc, gfx950): thin wrappers over the CDNA4 instructions.  tests/emu provides the same names for a
// host build (BF_HOST_EMU: one fibre per lane, collectives by cooperative scheduling) so that the kernel's control
// flow -- per-lane chain state machines, cross-wave reductions, barrier placement -- runs against the CPU oracle
// without a GPU.  The emulation is test infrastructure; nothing in the package loads it.
#pragma once
#ifdef BF_HOST_EMU
#include "emu_lane.h"
#else
#include <hip/hip_runtime.h>
#include "bfhip_model.h"

#define BF_DEV __device__ __forceinline__
typedef d4_t bf_acc4;
BF_DEV bf_acc4 bf_acc4_zero() { return bf_acc4{0., 0., 0., 0.}; }

BF_DEV int bf_tid() { return (int)threadIdx.x; }
BF_DEV int bf_group() { return (int)blockIdx.x; }
BF_DEV void bf_sync() { __syncthreads(); }
BF_DEV bool bf_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; acc[r] is
// D[4 r + (l >> 4)][l & 15].  Every entry accumulates as one sequential fma chain over k (tools/probe/mfma_arith_probe.hip).
BF_DEV bf_acc4 bf_mfma(double a, double b, bf_acc4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
// v + (the value of lane l ^ 16), v + (the value of lane l ^ 32): gfx950 row swaps
BF_DEV double bf_xor16_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
BF_DEV double bf_xor32_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// Two values reduced over lane pairs by ONE swap: v_permlane16_swap(a, b) leaves rows (a0, b0, a2, b2) in its first result
// and (a1, b1, a3, b3) in its second, so their sum is a + a(l ^ 16) in the even rows and b + b(l ^ 16) in the odd rows;
// v_permlane32_swap likewise for the lower / upper 32 lanes.  (The same additions as bf_xor16_add / bf_xor32_add.)
BF_DEV double bf_pair16_add(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
BF_DEV double bf_pair32_add(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// the value of lane l ^ 16.  v_permlane16_swap with both operands v leaves rows (0, 0, 2, 2) of v in the first result
// and rows (1, 1, 3, 3) in the second: a lane of an even row finds its partner in the second, of an odd row in the first
BF_DEV double bf_xor16_get(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const bool odd = (threadIdx.x >> 4) & 1;
    return __hiloint2double(odd ? b[0] : b[1], odd ? a[0] : a[1]);
}
BF_DEV double bf_exp(double x) { return exp(x); }
BF_DEV double bf_log(double x) { return log(x); }
BF_DEV double bf_sqrt(double x) { return sqrt(x); }
BF_DEV void bf_sincospi(double x, double *s, double *c) { sincospi(x, s, c); }
BF_DEV double bf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
BF_DEV double bf_fabs(double x) { return __builtin_fabs(x); }
BF_DEV void bf_atomic_add_u64(unsigned long long *p, unsigned long long v) { atomicAdd(p, v); }
#endif
