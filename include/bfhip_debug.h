/* bfhip_debug.h -- test and tuning switches of libbfhip.so.  NOT part of the drop-in boundary (include/bfhip.h): nothing a
 * caller of the library needs, nothing the host side of bayesfast_amd uses outside tests/, tools/ and bench.py's measurement
 * legs.  The switches are process-wide and not thread-safe; they select among kernels that give the same results (the tests
 * that use them compare exactly that), they never change what a call computes.
 *
 * Every switch also has an environment variable BFHIP_<KEY IN UPPER CASE> read once, when the library is first used
 * (bfhip_tune.h lists them with their meaning). */
#ifndef BFHIP_DEBUG_H
#define BFHIP_DEBUG_H
#include "bfhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Integer switches by name: "no_group", "no_pipe", "no_plain", "no_quad", "wave_cpg", "tail_relaunch", "tail_stop", "tail_q",
 * "tail_max", "lone", "lone_form", "pld_waves", "cubic_form", "cubic_loops", "gram_one_wave", "chol_one_panel", "no_vel_ahead", "tnuts_wpb", "tnuts_generic", "no_bound_proof", "no_proof_weights", "pld_no_compress", "pld_no_cl", "polar_tiles", "no_decay_shared", "no_group_pld".
 * Returns 0, or BFHIP_ERR_ARG for an unknown key. */
int bfhip_debug_set(const char *key, long long value);
/* The current value of a switch (0 for an unknown key). */
long long bfhip_debug_get(const char *key);
/* Device buffers the kernels write measurements into (NULL detaches): "stamps" (bf_sampler_kernel / bf_nuts_pipe_kernel cycle
 * stamps), "stamps_lone" (bf_lone_kernel), "gstamps" (group / split kernels), "group_counters" (4 x uint64: trips, trips with
 * the bound's tiles, with a late exchange, without the early one). */
int bfhip_debug_buffer(const char *key, void *device_ptr);
/* The kernel the last bfhip_sampler_run dispatched to. */
const char *bfhip_debug_last_kernel(void);
/* How many chains the last two-part launch listed for its second part (synchronises), -1 without one. */
int bfhip_debug_tail_count(bfhip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
