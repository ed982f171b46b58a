#include "bfhip_common.h"
extern "C" int bfhip_sampler_run(bfhip_ctx *ctx, const bfhip_sampler_config *cfg, int n_chain, int iter_end,
                                 uint64_t *rng, double *sc, double *vec, int iter_out0, int n_out, double *samples,
                                 double *stats, unsigned long long *n_leapfrog) {
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "sampler not built yet");
}
