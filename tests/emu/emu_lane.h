// emu_lane.h -- host emulation of the wave / workgroup primitives of bayesfast_amd/csrc/bfhip_lane.h.
// TEST INFRASTRUCTURE ONLY (tests/emu): one fibre (ucontext) per lane of a workgroup, scheduled cooperatively on one OS
// thread; a collective (barrier, row swap, ballot, MFMA) parks the calling fibre until every lane of its scope has
// arrived, so a collective called from divergent control flow deadlocks and is reported instead of passing silently.
// The arithmetic of v_mfma_f64_16x16x4_f64 is the sequential fma chain over k measured on gfx950
// (tools/probe/mfma_arith_probe.hip).
#pragma once
#include <cmath>
#include <cstdint>
#include "bfhip_model.h"

#define BF_DEV static inline

struct bf_acc4 {
    double v[4];
    double &operator[](int i) { return v[i]; }
    const double &operator[](int i) const { return v[i]; }
};
BF_DEV bf_acc4 bf_acc4_zero() { return bf_acc4{{0., 0., 0., 0.}}; }

int emu_tid();
int emu_group();
void emu_sync();
bool emu_any(bool p);
double emu_xor_add(double v, int mask);
double emu_xor_get(double v, int mask);
double emu_pair_add(double a, double b, int mask);
bf_acc4 emu_mfma(double a, double b, bf_acc4 c);

BF_DEV int bf_tid() { return emu_tid(); }
BF_DEV int bf_group() { return emu_group(); }
BF_DEV void bf_sync() { emu_sync(); }
BF_DEV bool bf_any(bool p) { return emu_any(p); }
BF_DEV bf_acc4 bf_mfma(double a, double b, bf_acc4 c) { return emu_mfma(a, b, c); }
BF_DEV double bf_xor16_add(double v) { return emu_xor_add(v, 16); }
BF_DEV double bf_xor32_add(double v) { return emu_xor_add(v, 32); }
BF_DEV double bf_xor16_get(double v) { return emu_xor_get(v, 16); }
BF_DEV double bf_pair16_add(double a, double b) { return emu_pair_add(a, b, 16); }
BF_DEV double bf_pair32_add(double a, double b) { return emu_pair_add(a, b, 32); }
BF_DEV double bf_exp(double x) { return std::exp(x); }
BF_DEV double bf_log(double x) { return std::log(x); }
BF_DEV double bf_sqrt(double x) { return std::sqrt(x); }
BF_DEV void bf_sincospi(double x, double *s, double *c) {
    const double t = 3.14159265358979323846 * x;
    *s = std::sin(t);
    *c = std::cos(t);
}
BF_DEV double bf_fma(double a, double b, double c) { return std::fma(a, b, c); }
BF_DEV double bf_fabs(double x) { return std::fabs(x); }
BF_DEV void bf_atomic_add_u64(unsigned long long *p, unsigned long long v) { *p += v; }
