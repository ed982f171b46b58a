// bfhip_group.hip -- device build and launch of the group sampler kernel (bfhip_group.h), gfx950.
// Compiled with -ffp-contract=off: the arithmetic is spelled out in the header (explicit fma where one is meant), so
// that a chain's numbers do not depend on what the optimiser fuses.
#include "bfhip_common.h"
#include "bfhip_group.h"
#include "bfhip_split.h"

// measurement hook (not part of include/bfhip.h): device buffer of two counters, trips and trips with the bound's tiles
static unsigned long long *g_gcount_ptr() { return bf_tune().group_counters; }
static unsigned long long *g_gstamps_ptr() { return bf_tune().gstamps; }
int bf_no_bound_proof() { return bf_tune().no_bound_proof; }

template <int W, bool NUTS, int FS>
__global__ __launch_bounds__(64 * W) void bf_group_kernel(DevModel m, SamplerArgs a) {
    extern __shared__ __attribute__((aligned(16))) double bf_group_lds[];
    bf_group_body<W, NUTS, FS>(m, a, bf_group_lds);
}

template <int W, bool NUTS, int FS>
static int launch_t(bfhip_ctx *ctx, const SamplerArgs &args) {
    auto k = bf_group_kernel<W, NUTS, FS>;
    size_t lds = GroupGeo<W>::lds_doubles((FS & 2) ? 3 : 2) * sizeof(double);
    if (FS & 8) {   // the pipeline density's block behind the group's own regions (sixteen-chain layout)
        const PldDev &pl = ctx->model.pld;
        lds = (((GroupGeo<W>::lds_doubles(((FS & 2) ? 3 : 2) - 1) + 1) & ~(size_t)1) + pld_lds_doubles(16 * W, pl.MP, pl.PP, 1, pl.n_ent, PLD_XS, W) +
               pld_cl_doubles(pl.MP, pl.PP) + 2) * sizeof(double);
    }
    if (lds > 64 * 1024)
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int groups = (args.n_chain + 15) / 16;
    hipLaunchKernelGGL(k, dim3(groups), dim3(64 * W), lds, ctx->stream, ctx->model, args);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int W>
static int launch_w(bfhip_ctx *ctx, const SamplerArgs &args, bool nuts, int fs) {
#ifdef BF_ONLY_HEADLINE  // tuning builds (tools/gvariant.sh): the 64-d plain NUTS instantiation only
    if (W == 4 && nuts && fs == 1) return launch_t<(W == 4 ? 4 : W), true, 1>(ctx, args);
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "tuning build: headline instantiation only");
#else
    if (nuts) {
        switch (fs) {
        case 1: return launch_t<W, true, 1>(ctx, args);
        case 3: return launch_t<W, true, 3>(ctx, args);
        case 5: return launch_t<W, true, 5>(ctx, args);
        case 7: return launch_t<W, true, 7>(ctx, args);
        case 8: return launch_t<W, true, 8>(ctx, args);     // the pipeline density (bfhip_group.h: PLDG)
        case 12: return launch_t<W, true, 12>(ctx, args);   // ... behind the constraint transform
        }
    } else {
        switch (fs) {
        case 1: return launch_t<W, false, 1>(ctx, args);
        case 3: return launch_t<W, false, 3>(ctx, args);
        case 5: return launch_t<W, false, 5>(ctx, args);
        case 7: return launch_t<W, false, 7>(ctx, args);
        }
    }
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "group kernel: feature set %d", fs);
#endif
}

// the split layout (bfhip_split.h): 2 W waves per 16 chains, W integrator and W bookkeeper waves -- two per SIMD at W = 4 (33 <= d
// <= 64), one per SIMD at W = 2 (17 <= d <= 32)
template <int W>
__global__ __launch_bounds__(128 * W) void bf_split_kernel(DevModel m, SamplerArgs a) {
    extern __shared__ __attribute__((aligned(16))) double bf_split_lds[];
    bf_split_body<true, W>(m, a, bf_split_lds);
}

bool bf_split_supports(const DevModel &m, const SamplerArgs &args) {
#ifdef BF_ONLY_HEADLINE
    if (m.DP != 64) return false;
#endif
    return m.DP <= 64 && args.cfg.sampler == 0 && m.has_quad && m.use_bound && !m.use_decay && !m.has_transform &&
           !m.has_su && !m.has_cubic && !m.has_link && !args.mat &&
           args.nslot >= (m.DP == 64 ? SplitGeoT<4>::scratch_slots() : (m.DP == 32 ? SplitGeoT<2>::scratch_slots() : SplitGeoT<1>::scratch_slots()));
}

int bf_launch_split(bfhip_ctx *ctx, const SamplerArgs &args_in) {
    SamplerArgs args = args_in;
    args.gcount = g_gcount_ptr();
    args.no_bound_proof = bf_no_bound_proof();
    args.stamps = g_gstamps_ptr();
    const int groups = (args.n_chain + 15) / 16;
#ifndef BF_ONLY_HEADLINE
    if (ctx->model.DP == 32) {
        const size_t lds = SplitGeoT<2>::lds_doubles() * sizeof(double);
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)bf_split_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bf_split_kernel<2>, dim3(groups), dim3(256), lds, ctx->stream, ctx->model, args);
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (ctx->model.DP == 16) {
        const size_t lds = SplitGeoT<1>::lds_doubles() * sizeof(double);
        BF_HIP_CHECK(hipFuncSetAttribute((const void *)bf_split_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bf_split_kernel<1>, dim3(groups), dim3(128), lds, ctx->stream, ctx->model, args);
        BF_HIP_CHECK(hipGetLastError());
        return 0;
    }
#endif
    const size_t lds = SplitGeoT<4>::lds_doubles() * sizeof(double);
    BF_HIP_CHECK(hipFuncSetAttribute((const void *)bf_split_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(bf_split_kernel<4>, dim3(groups), dim3(512), lds, ctx->stream, ctx->model, args);
    BF_HIP_CHECK(hipGetLastError());
    return 0;
}

// LDS bytes of the group kernel on the pipeline density (0: not a pipeline density)
static size_t group_pld_lds_bytes(const DevModel &m) {
    if (!m.pld.on) return 0;
    const int W = m.DP / 16;
    const size_t own = W == 4 ? GroupGeo<4>::lds_doubles(1) : (W == 2 ? GroupGeo<2>::lds_doubles(1) : GroupGeo<1>::lds_doubles(1));
    // (one operand region -- no decay term here --, K-split 1 of the second contraction and the row-major copy of C': bfhip_group.h)
    return (((own + 1) & ~(size_t)1) + pld_lds_doubles(m.DP, m.pld.MP, m.pld.PP, 1, m.pld.n_ent, PLD_XS, W) + pld_cl_doubles(m.pld.MP, m.pld.PP) + 2) * sizeof(double);
}

bool bf_group_supports(const DevModel &m, const SamplerArgs &args) {
    if (m.pld.on)   // the pipeline density (round 6): NUTS, no decay term, the sixteen-chain LDS layout has to fit beside the tree vectors
        return m.DP <= 64 && args.cfg.sampler == 0 && !m.use_decay && !args.mat && !(bf_tune().no_group_pld != 0) &&
               group_pld_lds_bytes(m) <= (size_t)160 * 1024;
    return m.DP <= 64 && m.has_quad && m.use_bound && !m.has_su && !m.has_cubic && !m.has_link && !args.mat;
}

int bf_group_scratch_slots(int DP) { return 5 * (BFHIP_MAX_TREEDEPTH - 2); }

// tuning hook (not part of include/bfhip.h): cycle stamps of workgroup 0's first trips, see GTRACE in bfhip_group.h

// test hook (not part of include/bfhip.h): 1 = never skip the bound's tiles (the results must not change)


int bf_launch_group(bfhip_ctx *ctx, const SamplerArgs &args_in) {
    SamplerArgs args = args_in;
    args.gcount = bf_tune().group_counters;
    args.stamps = bf_tune().gstamps;
    args.no_bound_proof = bf_tune().no_bound_proof;
    const DevModel &m = ctx->model;
    const bool nuts = args.cfg.sampler == 0;
    const int fs = m.pld.on ? (8 | (m.has_transform ? 4 : 0)) : (1 | (m.use_decay ? 2 : 0) | (m.has_transform ? 4 : 0));
    switch (m.DP / 16) {
#ifndef BF_ONLY_HEADLINE
    case 1: return launch_w<1>(ctx, args, nuts, fs);
    case 2: return launch_w<2>(ctx, args, nuts, fs);
#endif
    case 4: return launch_w<4>(ctx, args, nuts, fs);
    }
    return bf_set_error(BFHIP_ERR_UNSUPPORTED, "group kernel: padded dimension %d", m.DP);
}
