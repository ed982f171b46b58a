// compile the lone kernel alone to look at its registers (hipcc -I bayesfast_amd/csrc ...): includes what bfhip_sampler.hip includes before it
#include <type_traits>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "bfhip_eval.h"
#include "bfhip_metric.h"
#include "bfhip_pld.h"
#include "bfhip_oob.h"
__device__ __attribute__((noinline)) static double bf_exp_ni(double x) { return exp(x); }
__device__ __attribute__((noinline)) static double bf_log_ni(double x) { return log(x); }
__device__ __attribute__((noinline)) static double bf_sqrt_ni(double x) { return sqrt(x); }
__device__ __attribute__((noinline)) static void bf_sincospi_ni(double x, double *s, double *c) { sincospi(x, s, c); }
#define exp(x) bf_exp_ni(x)
#define log(x) bf_log_ni(x)
#define sqrt(x) bf_sqrt_ni(x)
#define sincospi(x, s, c) bf_sincospi_ni(x, s, c)
#define uexp(x) rfl(bf_exp_ni(x))
#define ulog(x) rfl(bf_log_ni(x))
#define usqrt(x) rfl(bf_sqrt_ni(x))
#include "bfhip_sampler_defs.h"
#include "bfhip_wave.h"
template <int W>
struct SamplerGeo {
    static constexpr int DP = 16 * W, NS = 4 * W;
    static constexpr int E = DP >= 64 ? DP / 64 : 1;
    static constexpr int XS = 65;
    static constexpr int GS = DP + 1;
    static constexpr int MAT = DP * DP;
    static constexpr bool STAGE = DP <= 64;
};
#define TRACE(k) do { } while (0)
#include "bfhip_nuts_pipe.h"
#include "bfhip_lone.h"
#ifndef LW
#define LW 2
#endif
#ifndef LMIN
#define LMIN 3
#endif
template __global__ void bf_lone_kernel<LW, false, false, LMIN>(DevModel, SamplerArgs);
