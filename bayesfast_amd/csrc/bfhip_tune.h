// bfhip_tune.h -- the library's test / tuning switches in one place (include/bfhip_debug.h is their interface).
// One process-wide instance (bfhip_api.hip), filled from the environment when first used.
#pragma once

struct BfTune {
    // --- kernel selection (all choices give the same results; tests compare them) ---
    int no_group;        // BFHIP_NO_GROUP / BFHIP_NUTS_KERNEL=sliced|pipe: keep NUTS / HMC off the lane-per-chain kernels
    int no_pipe;         // BFHIP_NO_PIPE / BFHIP_NUTS_KERNEL=sliced: NUTS on bf_sampler_kernel instead of the pipelined kernel
    int no_plain;        // BFHIP_NO_PLAIN: the run-time feature set instead of the compile-time instantiations
    int no_quad;         // BFHIP_NO_QUAD: 16-column tiles whatever the number of chains in a workgroup
    int wave_cpg;        // BFHIP_WAVE_CPG: chains per workgroup of the wave-per-chain kernels (0: automatic)
    int tail_relaunch;   // BFHIP_TAIL_RELAUNCH (default 1): the stragglers of a sixteen-chain launch in a second launch
    int tail_stop;       // BFHIP_TAIL_STOP (default 4): a workgroup with at most this many unfinished chains may stop them
    int tail_q;          // BFHIP_TAIL_Q (default 3): ... once this many quarters of the launch's chains are through
    int tail_max;        // BFHIP_TAIL_MAX (default 4): plain sliced kernel, chains of a group that may take the VALU matvec
    int lone;            // BFHIP_LONE (default 1): the latency kernel -- 1 automatic, 0 never, 2 wherever it is implemented
    int lone_form;       // BFHIP_LONE_FORM (default -1: by occupancy): 0 roomy, 1 three job waves (d > 32), 2 tight registers
    int pld_waves;       // BFHIP_PLD_WAVES: 8 or 16 waves per workgroup for the pipeline density (0: by chain count)
    int cubic_form;      // BFHIP_CUBIC_FORM (default 0: by chain count): d = 128 cubic surrogate -- 8 the eight-wave form, 4 four waves with S in registers
    int gram_one_wave;   // BFHIP_GRAM_ONE_WAVE: the Gram matrix with one wave per 64 x 64 block at every size (the same partial sums)
    int chol_one_panel;  // BFHIP_CHOL_ONE_PANEL: the Cholesky factorisation with one panel per pass over the trailing matrix at every size
    int cubic_loops;     // BFHIP_CUBIC_LOOPS: cubic configs by the general loops also at sixteen masked inputs (the same sums in the same order)
    int no_vel_ahead;    // BFHIP_NO_VEL_AHEAD: full-rank metric without the next step's velocity taken ahead
    int tnuts_wpb;       // BFHIP_TNUTS_WPB: tempered NUTS, chains per workgroup (4, 8; 0: automatic)
    int tnuts_generic;   // BFHIP_TNUTS_GENERIC: tempered NUTS on the generic kernel (bfhip_tnuts_gen.hip) also where the tuned one applies (tests compare them)
    int no_bound_proof;  // BFHIP_NO_BOUND_PROOF: always compute the H (x - mu) tiles
    int no_proof_weights;// BFHIP_NO_PROOF_WEIGHTS: the bound proof with the plain norm (read at upload)
    int no_group_pld;    // BFHIP_NO_GROUP_PLD: keep the pipeline density off the lane-per-chain group kernel (tests compare)
    int polar_tiles;     // BFHIP_POLAR_TILES: bfhip_polar_ns runs its first form (a tile per wave, two grid barriers per step) also at d <= 256 (tests compare)
    int no_decay_shared; // BFHIP_NO_DECAY_SHARED: the pipelined kernel runs the decay term's third matrix also when it is the bound's (tests compare)
    int pld_no_cl;       // BFHIP_PLD_NO_CL: the pipeline density's contractions stream their A fragments from L2 also where the LDS copy fits (tests compare)
    int pld_no_compress; // BFHIP_PLD_NO_COMPRESS: the pipeline density without the output-space compression (read at upload)
    // --- measurement buffers (device pointers or NULL) ---
    unsigned long long *stamps, *stamps_lone, *gstamps, *group_counters;
    char last_kernel[96];
};

BfTune &bf_tune();
