// bfhip_wave.h -- wave-level helpers of the wave-per-chain sampler kernels (gfx950): DPP moves, uniform-value hints, the
// 64-lane sums on the idle matrix pipe.  Included by bfhip_sampler.hip and bfhip_tnuts.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// one DPP move of a double (both halves)
template <int CTRL>
__device__ inline double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Wave-uniform values that arrive through a vector load (LDS, global memory) or an out-of-line call look
// divergent to the compiler, which then keeps the whole chain state machine in VGPRs and lowers its branches to
// exec-mask manipulation.  rfl() states the uniformity: the value moves to scalar registers and everything
// derived from it (unit, mode, depth, the branch conditions) stays scalar.
__device__ inline int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline double rfl(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ inline uint64_t rfl(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// gfx950 row swaps: every lane ends with (its row) + (the neighbouring row), then (its half) + (the other half)
__device__ inline double swap16_add_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ inline double swap32_add_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// Sums over the 64 lanes of N independent values, wave-uniform results in a fixed order.  The N reductions advance
// step by step together so that the latency of each step is covered by the other values' instructions.
// Default: two v_mfma_f64_4x4x4 per value (the matrix pipe is idle in the chain phase) and two row rotations; with
// BF_WSUM_BUTTERFLY: four DPP butterfly steps inside the rows of 16 lanes and two gfx950 row swaps (11 % slower on the
// headline workload, kept as the reference form of the reduction).
template <int N>
__device__ inline void wave_sum_n(double (&v)[N]) {
#ifndef BF_WSUM_BUTTERFLY
    // with B = 1 the first MFMA leaves, in lane 16i + 4b + j, the sum of the four lanes 16k + 4b + i (k = 0..3); fed
    // back as the B operand with A = 1 the second sums those over i: every lane of block b = (lane >> 2) & 3 holds the
    // total of its block's 16 lanes; two row rotations add the four blocks.
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[i], 1., 0., 0, 0, 0);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(1., v[i], 0., 0, 0, 0);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f64<0x128>(v[i]);  // row_ror:8
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f64<0x124>(v[i]);  // row_ror:4
#else
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f64<0xB1>(v[i]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f64<0x4E>(v[i]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f64<0x141>(v[i]);  // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_f64<0x140>(v[i]);  // row_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = swap16_add_f64(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = swap32_add_f64(v[i]);
#endif
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = rfl(v[i]);
}
__device__ inline double wave_sum(double v) {
    double t[1] = {v};
    wave_sum_n<1>(t);
    return t[0];
}

