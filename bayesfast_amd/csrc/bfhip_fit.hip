#include "bfhip_common.h"
extern "C" int bfhip_design_block(bfhip_ctx *ctx, int order, int n, int n_in, const double *x, const double *w,
                                  double *A, int lda, int col0) { return bf_set_error(BFHIP_ERR_UNSUPPORTED, "fit not built yet"); }
extern "C" int bfhip_gram(bfhip_ctx *ctx, int n, int P, int m, const double *A, int lda, const double *B, double *G, double *r) { return bf_set_error(BFHIP_ERR_UNSUPPORTED, "fit not built yet"); }
extern "C" int bfhip_solve_spd(bfhip_ctx *ctx, int P, int m, double *G, double *r, int *info) { return bf_set_error(BFHIP_ERR_UNSUPPORTED, "fit not built yet"); }
