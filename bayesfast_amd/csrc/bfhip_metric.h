// bfhip_metric.h -- full-rank metric (QuadMetricFull / QuadMetricFullAdapt, samplers/hmc_utils/metrics.py:94-132,
// 240-330, 374-417): per-chain d x d matrices and the wave-level routines on them.
//
// One wave owns one chain; lane l holds rows / dimensions l*E .. l*E+E-1 (E = 1 for d <= 64, 2 up to 128).
// Every matrix M of a chain is stored TRANSPOSED, MT[k * d + i] = M[i][k], so that for a fixed column k the
// rows of all lanes are one contiguous, coalesced segment and a lane only ever re-reads what it wrote itself;
// values of other rows travel by v_readlane.  The Cholesky factor also has a row-major copy (for the back
// substitution of metric.random, which walks rows).  Arithmetic is written with explicit non-fused multiplies and
// adds in the order of the CPU restatement (oracle/bf_oracle.c), so both sides round identically.
#pragma once
#include "bfhip_common.h"

enum { BF_MAT_COV = 0, BF_MAT_CHOL, BF_MAT_CHOL_ROWS, BF_MAT_FG, BF_MAT_BG, BF_MAT_WORK, BF_MAT_N };
static_assert(BF_MAT_N == BFHIP_MAT_N, "include/bfhip.h");

__device__ inline double bf_readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// element k of a vector held as x[e] of lane k / E (E <= 2)
template <int E>
__device__ inline double bf_pick(const double (&x)[E], int k) {
    return bf_readlane_f64((E > 1 && (k % E)) ? x[E - 1] : x[0], k / E);
}

// scipy.linalg.cholesky(a, lower=True): aT (transposed storage, only the lower triangle of a is read) -> lT.
// Returns false when a pivot is not positive (the reference keeps its previous factor then, metrics.py:287-292).
template <int E>
__device__ inline bool bf_chol_rows(const double *aT, double *lT, int d, int lane) {
    // Columns in blocks of JB: the finished columns k < j0 are read ONCE per block and applied to all its columns (column by
    // column every one of them was read again for every later column: 1 MB per factorisation at d = 64, the largest part of
    // a warm-up iteration with the full-rank metric), the block's own columns stay in registers.  For every column j the
    // terms L[i][k] L[j][k] are subtracted for k = 0 .. j - 1 in this order, as before (oracle/bf_oracle.c rounds the same).
    // (Writing the row-major copy from the block's registers, whole cache lines into a second work matrix, and publishing by
    // two straight copies was measured slower than bf_chol_publish's scattered stores: 262 against 219 ms per 100 iterations.)
    constexpr int JB = E == 1 ? 16 : 8, KB = 8;
    for (int j0 = 0; j0 < d; j0 += JB) {
        const int nj = d - j0 < JB ? d - j0 : JB;
        double acc[JB][E];
#pragma unroll
        for (int u = 0; u < JB; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                acc[u][e] = (u < nj && i < d) ? aT[(size_t)(j0 + u) * d + i] : 0.;  // a[i][j0 + u]
            }
        for (int k = 0; k < j0; k += KB) {  // (j0 is a multiple of JB, JB of KB)
            double lb[KB][E];
#pragma unroll
            for (int v = 0; v < KB; ++v)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = lane * E + e;
                    lb[v][e] = (i < d) ? lT[(size_t)(k + v) * d + i] : 0.;  // L[i][k + v], written by this lane earlier
                }
#pragma unroll
            for (int v = 0; v < KB; ++v)
#pragma unroll
                for (int u = 0; u < JB; ++u)
                    if (u < nj) {
                        const double ljk = bf_pick<E>(lb[v], j0 + u);              // L[j0 + u][k + v]
#pragma unroll
                        for (int e = 0; e < E; ++e) acc[u][e] = __dsub_rn(acc[u][e], __dmul_rn(lb[v][e], ljk));
                    }
        }
#pragma unroll
        for (int u = 0; u < JB; ++u) {
            if (u < nj) {
                const int j = j0 + u;
#pragma unroll
                for (int v = 0; v < u; ++v) {   // the block's own finished columns, from registers
                    const double ljk = bf_pick<E>(acc[v], j);                      // L[j][j0 + v]
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[u][e] = __dsub_rn(acc[u][e], __dmul_rn(acc[v][e], ljk));
                }
                const double s = bf_pick<E>(acc[u], j);
                if (!(s > 0.)) return false;
                const double ljj = __dsqrt_rn(s);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = lane * E + e;
                    acc[u][e] = i > j ? __ddiv_rn(acc[u][e], ljj) : (i == j ? ljj : 0.);
                    if (i < d) lT[(size_t)j * d + i] = acc[u][e];
                }
            }
        }
    }
    return true;
}

// lT -> the two stored copies of the factor
template <int E>
__device__ inline void bf_chol_publish(const double *lT, double *cholT, double *cholR, int d, int lane) {
    // columns in batches: the loads of a batch before its stores (the matrices of a chain may alias for all the compiler
    // knows, so a load -> store loop waits a memory latency per column)
    constexpr int B = 16;
    int j = 0;
    for (; j + B <= d; j += B) {
        double v[B][E];
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                v[u][e] = (i < d) ? lT[(size_t)(j + u) * d + i] : 0.;
            }
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                if (i < d) {
                    cholT[(size_t)(j + u) * d + i] = v[u][e];
                    cholR[(size_t)i * d + j + u] = v[u][e];  // row-major: L[i][j + u]
                }
            }
    }
    for (; j < d; ++j) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            if (i < d) {
                const double v = lT[(size_t)j * d + i];
                cholT[(size_t)j * d + i] = v;
                cholR[(size_t)i * d + j] = v;
            }
        }
    }
}

// velocity = cov p (np.dot(cov, x), metrics.py:113-115): out_i = sum_k cov[i][k] p_k, k ascending
template <int E>
__device__ inline void bf_velocity_full(const double *covT, const double (&pv)[E], double (&out)[E], int d, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = 0.;
    // columns in blocks of BF_MV_BLOCK: all loads of a block are in flight before the first multiply-add (a dependent
    // load -> add chain per column leaves the wave waiting a full memory latency 64 times); the sum stays k-ascending
    constexpr int B = 16;
    int k = 0;
    for (; k + B <= d; k += B) {
        double c[B][E];
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                c[u][e] = (i < d) ? covT[(size_t)(k + u) * d + i] : 0.;
            }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const double pk = bf_pick<E>(pv, k + u);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                if (i < d) out[e] = __dadd_rn(out[e], __dmul_rn(c[u][e], pk));
            }
        }
    }
    for (; k < d; ++k) {
        const double pk = bf_pick<E>(pv, k);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            if (i < d) out[e] = __dadd_rn(out[e], __dmul_rn(covT[(size_t)k * d + i], pk));
        }
    }
}

// Two velocities in ONE pass over the matrix: a leapfrog step takes cov p' at its end (integration.py:92) and cov (p' + eps/2 g') at
// the start of the next one (:82) -- both vectors are known when the first product is taken, and the 32 KB of a chain's
// covariance are what the full-rank metric's kernel is bound by.  Each sum is bf_velocity_full's, term by term.
template <int E>
__device__ inline void bf_velocity_full2(const double *covT, const double (&p0)[E], const double (&p1)[E], double (&o0)[E], double (&o1)[E],
                                         int d, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) o0[e] = o1[e] = 0.;
    constexpr int B = 16;
    int k = 0;
    for (; k + B <= d; k += B) {
        double c[B][E];
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                c[u][e] = (i < d) ? covT[(size_t)(k + u) * d + i] : 0.;
            }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const double a0 = bf_pick<E>(p0, k + u), a1 = bf_pick<E>(p1, k + u);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                if (i < d) {
                    o0[e] = __dadd_rn(o0[e], __dmul_rn(c[u][e], a0));
                    o1[e] = __dadd_rn(o1[e], __dmul_rn(c[u][e], a1));
                }
            }
        }
    }
    for (; k < d; ++k) {
        const double a0 = bf_pick<E>(p0, k), a1 = bf_pick<E>(p1, k);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            if (i < d) {
                const double cv = covT[(size_t)k * d + i];
                o0[e] = __dadd_rn(o0[e], __dmul_rn(cv, a0));
                o1[e] = __dadd_rn(o1[e], __dmul_rn(cv, a1));
            }
        }
    }
}

// Three velocities in ONE pass over the matrix (the U-turn checks of a merge level and of a doubling's end need cov p of three
// stored momenta: nuts.py:150-160, 88-100): each vector's sum is the one bf_velocity_full makes, term by term; the 32 KB of a
// chain's covariance are read once instead of three times.
template <int E>
__device__ inline void bf_velocity_full3(const double *covT, const double (&p0)[E], const double (&p1)[E], const double (&p2)[E],
                                         double (&o0)[E], double (&o1)[E], double (&o2)[E], int d, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) o0[e] = o1[e] = o2[e] = 0.;
    constexpr int B = 16;
    int k = 0;
    for (; k + B <= d; k += B) {
        double c[B][E];
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                c[u][e] = (i < d) ? covT[(size_t)(k + u) * d + i] : 0.;
            }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const double a0 = bf_pick<E>(p0, k + u), a1 = bf_pick<E>(p1, k + u), a2 = bf_pick<E>(p2, k + u);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                if (i < d) {
                    o0[e] = __dadd_rn(o0[e], __dmul_rn(c[u][e], a0));
                    o1[e] = __dadd_rn(o1[e], __dmul_rn(c[u][e], a1));
                    o2[e] = __dadd_rn(o2[e], __dmul_rn(c[u][e], a2));
                }
            }
        }
    }
    for (; k < d; ++k) {
        const double a0 = bf_pick<E>(p0, k), a1 = bf_pick<E>(p1, k), a2 = bf_pick<E>(p2, k);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            if (i < d) {
                const double cv = covT[(size_t)k * d + i];
                o0[e] = __dadd_rn(o0[e], __dmul_rn(cv, a0));
                o1[e] = __dadd_rn(o1[e], __dmul_rn(cv, a1));
                o2[e] = __dadd_rn(o2[e], __dmul_rn(cv, a2));
            }
        }
    }
}

// metric.random for the full metric: solve L^T p = z (scipy.linalg.solve_triangular(chol.T, z), metrics.py:123-127)
// by the column sweep of BLAS dtrsv: j = d-1 .. 0: p_j = z_j / L[j][j]; z_i -= L[j][i] p_j for i < j.
template <int E>
__device__ inline void bf_solve_lt(const double *cholR, double (&z)[E], int d, int lane) {
    constexpr int B = 8;
    int j = d - 1;
    for (; j - B + 1 >= 0; j -= B) {
        double rb[B][E];
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                rb[u][e] = (i <= j - u) ? cholR[(size_t)(j - u) * d + i] : 0.;  // L[j - u][i]
            }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int jj = j - u;
            const double pj = __ddiv_rn(bf_pick<E>(z, jj), bf_pick<E>(rb[u], jj));
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                if (i < jj) z[e] = __dsub_rn(z[e], __dmul_rn(rb[u][e], pj));
                else if (i == jj) z[e] = pj;
            }
        }
    }
    for (; j >= 0; --j) {
        double row[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            row[e] = (i <= j) ? cholR[(size_t)j * d + i] : 0.;  // L[j][i]
        }
        const double pj = __ddiv_rn(bf_pick<E>(z, j), bf_pick<E>(row, j));
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            if (i < j) z[e] = __dsub_rn(z[e], __dmul_rn(row[e], pj));
            else if (i == j) z[e] = pj;
        }
    }
}

// _WeightedCovariance.add_sample (metrics.py:401-407) on the transposed accumulator: raw[i][j] += new_i * old_j
template <int E>
// (scaledT: when not NULL also scaledT = updated raw / n -- _update_from_weightvar's covariance, metrics.py:287-292, taken
// while the accumulator passes through registers instead of in a second sweep over it)
__device__ inline void bf_welford_cov(double *rawT, const double (&new_diff)[E], const double (&old_diff)[E], int d, int lane,
                                      double *scaledT = nullptr, double n = 1.) {
    constexpr int B = 8;
    int j = 0;
    for (; j + B <= d; j += B) {
        double rb[B][E];
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                rb[u][e] = (i < d) ? rawT[(size_t)(j + u) * d + i] : 0.;
            }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const double oj = bf_pick<E>(old_diff, j + u);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                if (i < d) {
                    const double r = __dadd_rn(rb[u][e], __dmul_rn(__dmul_rn(1., new_diff[e]), oj));
                    rawT[(size_t)(j + u) * d + i] = r;
                    if (scaledT) scaledT[(size_t)(j + u) * d + i] = r / n;
                }
            }
        }
    }
    for (; j < d; ++j) {
        const double oj = bf_pick<E>(old_diff, j);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            if (i < d) {
                const double r = __dadd_rn(rawT[(size_t)j * d + i], __dmul_rn(__dmul_rn(1., new_diff[e]), oj));
                rawT[(size_t)j * d + i] = r;
                if (scaledT) scaledT[(size_t)j * d + i] = r / n;
            }
        }
    }
}
