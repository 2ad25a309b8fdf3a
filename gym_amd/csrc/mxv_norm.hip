// mxv_norm.hip — SURVEY.md §8(f)-2: gym.wrappers.NormalizeObservation / NormalizeReward
// (gym/wrappers/normalize.py:8-144) for a vector env, as gfx950 kernels behind the mxv_norm_* C ABI (include/mxv.h).
//
// The reference normalises one batch per step() call: RunningMeanStd.update(batch) (:17-29, a Chan/Welford merge of
// the batch mean/var into the running mean/var, :32-47) followed by an element-wise affine map (:90-93, :143-145).
// The batch moments are the only cross-env reduction on the whole hot path.  Here a chunk of K consecutive batches
// ([K][N][O] trajectory tensors written by mxv_rollout) is processed by four launches, all HBM-streaming:
//
//   1. *_sums_kernel     per step k and per leaf (4096 rows of observations / 256 envs of returns): fp64 sum and sum of
//                        squares of every column.  NormalizeReward's discounted-return recurrence (:132,:136) runs here
//                        too, one env per lane, K steps sequentially, so the returns never touch HBM.
//   2. tree_kernel       fixed binary tree over the leaves (by leaf index) -> per-step sums of this shard.  A fixed tree
//                        makes the result independent of scheduling AND of how a logical vector env is sharded over
//                        1/2/4/8 GPUs (power-of-two shards of >= 4096 rows are complete subtrees of the same tree).
//   3. scan_kernel       one thread per column walks the K steps: combines the shards' sums (tree over ranks), forms the
//                        batch mean/var IN THE REFERENCE'S MOMENT DTYPE (float32 for observations — np.mean/np.var of a
//                        float32 array — float64 for returns), applies update_mean_var_count_from_moments in source
//                        order, and emits (mean_k, sqrt(var_k + epsilon)) per step.
//   4. *_apply_kernel    y = (x - mean_k) / denom_k  resp.  r / denom_k, IEEE fp64 subtraction and division as in the
//                        reference (float32 - float64 -> float64).
//
// Numerics: sums are exact-order fp64 (no atomics), so results are bit-reproducible; they differ from the reference only
// by the float32 accumulation error the reference's own np.mean/np.var carry (the test suite's CPU restatement has both
// arithmetics: the reference's, pinned bit-exact by goldens generated from the live wrappers, and this definition).
// Built with -ffp-contract=off like the rest of the engine: explicit __fma_rn only where the product is exact anyway.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdarg>
#include <cstdio>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mxv.h"

namespace {

constexpr int kThreads = 256;
constexpr int kObsLeafRows = 4096;  // rows of observations per leaf of the sum tree
constexpr int kRewLeafEnvs = 256;   // envs per leaf of the return-sum tree (one wave64, 4 envs per lane)
constexpr int kTreeFan = 1024;      // leaves folded per workgroup per tree level (a complete 10-level subtree)
constexpr int kMaxWorld = 64;

#ifndef MXV_NORM_DPP_REDUCE
#define MXV_NORM_DPP_REDUCE 1  // A/B hook: 0 = the six ds_bpermute stages of __shfl_xor
#endif
#ifndef MXV_NORM_XCD_MAP
#define MXV_NORM_XCD_MAP 1     // A/B hook: 0 = leaf = workgroup id
#endif
#ifndef MXV_NORM_VEC4
#define MXV_NORM_VEC4 1        // A/B hook: 0 = lane L of a return leaf owns envs L, L + 64, L + 128, L + 192 (twelve loads per step)
#endif

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_tree_sum(double v) {
    // xor butterfly = the binary tree over lane index (addition is commutative, so every lane holds the tree's value)
#if MXV_NORM_DPP_REDUCE
    // The same tree without LDS round trips.  Stages 1, 2: quad permutes (lane ^ 1, lane ^ 2).  Stages 4, 8: after them every lane of
    // a quad (of a half row) holds the same partial sum, so the mirror permutes of the DPP unit — lane 7 - i of the half row, lane
    // 15 - i of the row — deliver what lane ^ 4 (lane ^ 8) holds.  Stages 16, 32: gfx950's v_permlane16_swap / v_permlane32_swap.
    // a + b == b + a exactly, so the value is the __shfl_xor butterfly's, bit for bit.
    v += dpp_f64<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);   // rows (0, 1), (2, 3)
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);   // lower half + upper half
    }
#else
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
#endif
    return v;
}

template <int O>
__device__ __forceinline__ void load_row(const float *__restrict__ p, float (&f)[O]) {
    if constexpr (O == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else if constexpr (O == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(p);
        f[0] = v.x; f[1] = v.y;
    } else if constexpr (O == 6) {
        const float2 a = reinterpret_cast<const float2 *>(p)[0], b = reinterpret_cast<const float2 *>(p)[1],
                     c = reinterpret_cast<const float2 *>(p)[2];
        f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = c.x; f[5] = c.y;
    } else {
#pragma unroll
        for (int j = 0; j < O; ++j) f[j] = p[j];
    }
}

// ---- 1a. observations: per-(step, leaf) column sums ---------------------------------------------------------------
// grid (leaves, K); partials [K][leaves][2*O] = (sum_0..sum_{O-1}, sumsq_0..sumsq_{O-1}).
template <int O>
__global__ void __launch_bounds__(kThreads) obs_sums_kernel(const float *__restrict__ x, int64_t n, int64_t leaves,
                                                            double *__restrict__ partials) {
    const int tid = threadIdx.x;
    const int64_t k = blockIdx.y, leaf = blockIdx.x;
    const int64_t row0 = leaf * kObsLeafRows;
    const int rows = (int)((n - row0) < kObsLeafRows ? (n - row0) : kObsLeafRows);
    const float *__restrict__ base = x + (k * n + row0) * O;
    double s[O], q[O];
#pragma unroll
    for (int j = 0; j < O; ++j) s[j] = q[j] = 0.0;
    // a lane's rows are tid, tid+256, ...: every wave load is a dense burst (64 rows x 4*O bytes)
#pragma unroll 4   // (8 or 16 rows in flight per lane measured the same: profiles/r4/r4n_normalize_variants.txt)
    for (int r = tid; r < rows; r += kThreads) {
        float f[O];
        load_row<O>(base + (int64_t)r * O, f);
#pragma unroll
        for (int j = 0; j < O; ++j) {
            const double v = (double)f[j];
            s[j] += v;
            q[j] = __fma_rn(v, v, q[j]);  // v*v is exact in fp64 (24-bit significands): one rounding, as mul+add would give
        }
    }
    __shared__ double sm[kThreads / 64][2 * O];
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < O; ++j) {
        const double a = wave_tree_sum(s[j]), b = wave_tree_sum(q[j]);
        if (lane == 0) {
            sm[wave][j] = a;
            sm[wave][O + j] = b;
        }
    }
    __syncthreads();
    if (tid < 2 * O)
        partials[(k * leaves + leaf) * (2 * O) + tid] = (sm[0][tid] + sm[1][tid]) + (sm[2][tid] + sm[3][tid]);
}

// ---- 1b. rewards: discounted-return recurrence + per-(step, leaf) sums ----------------------------------------------
// One wave per workgroup, leaf = 256 consecutive envs, lane L owns envs leaf*256 + 4*L + j (j < 4): per step one 32-byte (float32
// rewards: 16-byte) load of rewards and one dword of each flag array per lane — four load instructions per wave-step where a lane
// that owned envs L, L + 64, ... issued twelve, eight of them 64-byte bursts of flag bytes.  (Vector loads need n % 4 == 0: the
// steps of a [K][n] tensor then start on multiples of four elements; other sizes load element by element, same lanes, same order.)
// partials [K][leaves][2] = (sum, sumsq) of the returns AFTER the update of step k and BEFORE the zeroing of finished envs
// (normalize.py:132-136).
template <typename RT>
__global__ void __launch_bounds__(64) returns_sums_kernel(const RT *__restrict__ rew, const uint8_t *__restrict__ term,
                                                          const uint8_t *__restrict__ trunc, double *__restrict__ returns,
                                                          int64_t n, int K, double gamma, int64_t leaves,
                                                          double *__restrict__ partials, int aligned) {
    const int lane = threadIdx.x;
#if MXV_NORM_XCD_MAP
    // workgroup ids are dealt round-robin over the 8 XCDs: XCD x takes the x-th contiguous eighth of the leaves (as the rollout kernels'
    // tiles), so that the lines one L2 fetches at a step are neighbours in memory
    int64_t leaf;
    {
        const unsigned bid = blockIdx.x, nb = gridDim.x, x = bid % 8u, idx = bid / 8u, base = nb / 8u, rem = nb % 8u;
        leaf = (int64_t)(x * base + (x < rem ? x : rem) + idx);
    }
#else
    const int64_t leaf = blockIdx.x;
#endif
    int64_t e[4];
    bool live[4];
    double ret[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        e[j] = MXV_NORM_VEC4 ? leaf * kRewLeafEnvs + 4 * lane + j : leaf * kRewLeafEnvs + j * 64 + lane;
        live[j] = e[j] < n;
        ret[j] = live[j] ? returns[e[j]] : 0.0;
    }
    // software pipeline: the loads of steps k+1 .. k+D-1 are in flight while step k is reduced (the recurrence itself is a few
    // instructions per step: without a prefetch every step would wait a full HBM round trip; with one step of look-ahead — rounds 1-3 —
    // the four waves of a SIMD still spent most of a step waiting: 3.5 us per 2^20-env step for 10 B per env-step = 3 TB/s).  A ring
    // of D register sets, the loop unrolled by D so that every set has a fixed name; the arithmetic and its order are unchanged.
    constexpr int D = 4;
    // A ring entry holds what the loads RETURN (rewards, the two flag words or bytes) and nothing derived from it: OR-ing the flags at
    // fetch time — rounds 1-4a — made every fetch wait for its own flag loads, a full HBM round trip per step with the "prefetched"
    // rewards of the later steps queued behind it (3.5 TB/s for 10 B per env-step).
    struct Slot {
        RT r[4];
        uint32_t ft[4], fu[4];   // VEC: element 0 = the dword of four flag bytes; else one byte each
    };
    Slot ring[D];
    // wave-uniform: the whole leaf exists and every step of the [K][n] tensors starts on a multiple of four elements
    const bool vec = MXV_NORM_VEC4 && aligned && (n & 3) == 0 && (leaf + 1) * kRewLeafEnvs <= n;   // aligned: the host checked the tensors' addresses
    auto run = [&](auto vec_tag) __attribute__((always_inline)) {
        constexpr bool VEC = decltype(vec_tag)::value;
        auto fetch = [&](int k, Slot &sl) {
            const int64_t off = (int64_t)(k < K ? k : K - 1) * n;   // past the end: the last step once more (never used)
            if constexpr (VEC) {
                if constexpr (sizeof(RT) == 8) {
                    const double2 a = reinterpret_cast<const double2 *>(rew + off + e[0])[0], b = reinterpret_cast<const double2 *>(rew + off + e[0])[1];
                    sl.r[0] = a.x; sl.r[1] = a.y; sl.r[2] = b.x; sl.r[3] = b.y;
                } else {
                    const float4 a = *reinterpret_cast<const float4 *>(rew + off + e[0]);
                    sl.r[0] = a.x; sl.r[1] = a.y; sl.r[2] = a.z; sl.r[3] = a.w;
                }
                sl.ft[0] = *reinterpret_cast<const uint32_t *>(term + off + e[0]);
                sl.fu[0] = *reinterpret_cast<const uint32_t *>(trunc + off + e[0]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sl.r[j] = live[j] ? rew[off + e[j]] : (RT)0;
                    sl.ft[j] = live[j] ? (uint32_t)term[off + e[j]] : 0u;
                    sl.fu[j] = live[j] ? (uint32_t)trunc[off + e[j]] : 0u;
                }
            }
        };
#pragma unroll
        for (int d = 0; d < D - 1; ++d) fetch(d, ring[d]);
        auto one_step = [&](int k, const Slot &sl) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (VEC || live[j]) {
                    ret[j] = ret[j] * gamma + (double)sl.r[j];  // :132 (two roundings; contraction is off)
                    s += ret[j];
                    q = __fma_rn(ret[j], ret[j], q);
                }
            }
            s = wave_tree_sum(s);
            q = wave_tree_sum(q);
            if (lane == 0) {
                partials[((int64_t)k * leaves + leaf) * 2 + 0] = s;
                partials[((int64_t)k * leaves + leaf) * 2 + 1] = q;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // :135-136
                const bool done = VEC ? (((sl.ft[0] | sl.fu[0]) >> (8 * j)) & 0xffu) != 0 : (sl.ft[j] | sl.fu[j]) != 0;
                if (done) ret[j] = 0.0;
            }
        };
        // full groups of D steps: a body without exits (a `break` between the unrolled steps splits it into blocks, and the waits the
        // compiler then places at the joins drain the ring once per group); the last K % D steps find their slots already loaded
        int k0 = 0;
        for (; k0 + D <= K; k0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                fetch(k0 + d + D - 1, ring[(d + D - 1) % D]);   // the slot step k-1 has just released
                one_step(k0 + d, ring[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (k0 + d < K) one_step(k0 + d, ring[d]);
    };
    if (vec) run(std::true_type{});
    else run(std::false_type{});
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (live[j]) returns[e[j]] = ret[j];
}

// ---- 2. one level of the fixed binary tree: in [K][leaves][V] -> out [K][ceil(leaves/1024)][V] ------------------------
// Missing leaves count as +0.0 (x + 0.0 == x): an odd leftover passes through unchanged, as a pairwise tree does.
__global__ void __launch_bounds__(kThreads) tree_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                        int64_t leaves, int V) {
    constexpr int kMaxV = 12;  // 2 * dim, dim <= 6
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t k = blockIdx.y, group = blockIdx.x, groups = gridDim.x;
    const int64_t l0 = group * kTreeFan + (int64_t)tid * 4;
    __shared__ double sm[kThreads / 64][kMaxV];
    // all V values of a lane's four leaves in one go (4 V contiguous doubles), ONE barrier per workgroup: the per-value loop with two
    // barriers each cost the fused-moments path (8192 leaves of 8 values per step) more than the scan; same tree, same bits
#pragma unroll
    for (int v = 0; v < kMaxV; ++v) {
        if (v < V) {
            double a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = (l0 + i < leaves) ? in[(k * leaves + l0 + i) * V + v] : 0.0;
            const double t = wave_tree_sum((a[0] + a[1]) + (a[2] + a[3]));
            if (lane == 0) sm[wave][v] = t;
        }
    }
    __syncthreads();
    if (tid < V) out[(k * groups + group) * V + tid] = (sm[0][tid] + sm[1][tid]) + (sm[2][tid] + sm[3][tid]);
}

// ---- 3. the running update, sequential over the K steps (normalize.py:17-47) ------------------------------------------
// all_sums [W][K][2*O]; stat = mean[O], var[O], count; coef [K][O][2] = (mean_k, sqrt(var_k + epsilon)).
// One workgroup, chunks of kScanChunk steps, three phases per chunk:
//   A (all threads, one (step, column) pair each): the tree over the ranks' sums and the batch moments — nothing here depends on the
//     running statistics — into LDS;
//   B (thread j = column j): the Chan merge itself, the only sequential part: three divisions per step, operands read from LDS
//     (addresses known in advance: the reads of the next steps are in flight while a step's divisions run);
//   C (all threads): coef_k = (mean_k, sqrt(var_k + epsilon)) out to HBM.
// Rounds 1-4a walked the steps with one thread per column doing everything: a dependent HBM round trip, the moments' divisions and a
// square root per step — 84 us for 128 steps (8 % of a NormalizeReward pass).  Every value is computed by the same operations in the
// same order as before.
constexpr int kScanChunk = 256, kScanMaxDim = 6;
__global__ void __launch_bounds__(kThreads) scan_kernel(const double *__restrict__ all_sums, int W, int K, int O, double total_rows,
                                                        double epsilon, int obs_dtype_f32, double *__restrict__ stat,
                                                        double *__restrict__ coef) {
    __shared__ double l_bm[kScanChunk * kScanMaxDim], l_mb[kScanChunk * kScanMaxDim];   // batch mean, m_b; then mean_k, var_k
    const int tid = threadIdx.x;
    const double N = total_rows;
    double mean = 0.0, var = 0.0, count = 0.0;
    if (tid < O) {
        mean = stat[tid];
        var = stat[O + tid];
        count = stat[2 * O];
    }
    for (int c0 = 0; c0 < K; c0 += kScanChunk) {
        const int steps = (K - c0) < kScanChunk ? (K - c0) : kScanChunk;
        for (int i = tid; i < steps * O; i += kThreads) {  // ---- A
            const int k = c0 + i / O, j = i % O;
            double bs[kMaxWorld], bq[kMaxWorld];
            for (int w = 0; w < W; ++w) {
                bs[w] = all_sums[((int64_t)w * K + k) * (2 * O) + j];
                bq[w] = all_sums[((int64_t)w * K + k) * (2 * O) + O + j];
            }
            for (int stride = 1; stride < W; stride <<= 1)  // binary tree over the rank index
                for (int w = 0; w + stride < W; w += 2 * stride) {
                    bs[w] += bs[w + stride];
                    bq[w] += bq[w + stride];
                }
            const double S = bs[0], Q = bq[0];
            double batch_mean, m_b;
            if (obs_dtype_f32) {
                // np.mean / np.var of a float32 array are float32 (the division itself runs in double: _methods.py _mean/_var);
                // var = mean((x - mean32)^2) expanded over the exact sums
                const float mean32 = (float)(S / N);
                const double m = (double)mean32;
                double v = ((Q - 2.0 * m * S) + N * m * m) / N;
                v = v > 0.0 ? v : 0.0;
                const float var32 = (float)v;
                batch_mean = m;
                m_b = (double)__fmul_rn(var32, (float)N);  // float32 array * python int stays float32 (:41)
            } else {
                batch_mean = S / N;
                double v = Q / N - batch_mean * batch_mean;
                v = v > 0.0 ? v : 0.0;
                m_b = v * N;
            }
            l_bm[i] = batch_mean;
            l_mb[i] = m_b;
        }
        __syncthreads();
        if (tid < O) {  // ---- B
#pragma unroll 4
            for (int kk = 0; kk < steps; ++kk) {
                const double batch_mean = l_bm[kk * O + tid], m_b = l_mb[kk * O + tid];
                const double delta = batch_mean - mean;                                  // :36
                const double tot = count + N;                                            // :37
                const double new_mean = mean + delta * N / tot;                          // :39
                const double m_a = var * count;                                          // :40
                const double M2 = m_a + m_b + delta * delta * count * N / tot;           // :42
                mean = new_mean;
                var = M2 / tot;                                                          // :43
                count = tot;                                                             // :44
                l_bm[kk * O + tid] = mean;
                l_mb[kk * O + tid] = var;
            }
        }
        __syncthreads();
        for (int i = tid; i < steps * O; i += kThreads) {  // ---- C
            coef[((int64_t)c0 * O + i) * 2 + 0] = l_bm[i];
            coef[((int64_t)c0 * O + i) * 2 + 1] = sqrt(l_mb[i] + epsilon);             // np.sqrt(var + epsilon), :93 / :145
        }
        __syncthreads();
    }
    if (tid < O) {
        stat[tid] = mean;
        stat[O + tid] = var;
        if (tid == 0) stat[2 * O] = count;
    }
}

// ---- 4a. (obs - mean) / sqrt(var + epsilon): one row per lane, grid (ceil(N/256), K) -----------------------------------
template <int O, typename OUT>
__global__ void __launch_bounds__(kThreads) obs_apply_kernel(const float *__restrict__ x, OUT *__restrict__ y,
                                                             const double *__restrict__ coef, int64_t n) {
    const int64_t k = blockIdx.y;
    const int64_t row = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (row >= n) return;
    const double *__restrict__ c = coef + k * O * 2;  // uniform per workgroup: scalar loads
    float f[O];
    load_row<O>(x + (k * n + row) * O, f);
    OUT o[O];
#pragma unroll
    for (int j = 0; j < O; ++j) o[j] = (OUT)(((double)f[j] - c[2 * j]) / c[2 * j + 1]);
    OUT *__restrict__ dst = y + (k * n + row) * O;
    if constexpr (O == 4 && sizeof(OUT) == 4) {
        *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else if constexpr (O % 2 == 0 && sizeof(OUT) == 8) {
#pragma unroll
        for (int j = 0; j < O; j += 2) reinterpret_cast<double2 *>(dst)[j / 2] = make_double2(o[j], o[j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < O; ++j) dst[j] = o[j];
    }
}

// ---- 4b. rews / sqrt(var + epsilon) -------------------------------------------------------------------------------------
// V elements per lane: 16-byte loads and stores (double2 / float4) when every step's slice starts 16-byte aligned (n % V == 0 and
// aligned tensors: the host checks), else one element per lane.  One element per lane kept 8 (4) bytes in flight per lane: 5.6 TB/s of
// read + write where the observation map's 16-byte lanes reach 6.2.
template <typename RT, int V>
__global__ void __launch_bounds__(kThreads) reward_apply_kernel(const RT *__restrict__ rew, RT *__restrict__ out,
                                                                const double *__restrict__ coef, int64_t n) {
    const int64_t k = blockIdx.y;
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * V;
    if (i >= n) return;
    const double denom = coef[k * 2 + 1];
    if constexpr (V == 1) {
        out[k * n + i] = (RT)((double)rew[k * n + i] / denom);
    } else {
        struct alignas(16) Pack { RT v[V]; };
        const Pack in = *reinterpret_cast<const Pack *>(rew + k * n + i);
        Pack o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = (RT)((double)in.v[j] / denom);
        *reinterpret_cast<Pack *>(out + k * n + i) = o;
    }
}

}  // namespace

struct mxv_norm {
    int device = 0, dim = 0;
    int64_t n = 0;
    hipStream_t stream = nullptr;
    double *stat = nullptr;     // mean[dim], var[dim], count
    double *returns = nullptr;  // [n] discounted returns of NormalizeReward (zeros until used)
    double *part_a = nullptr, *part_b = nullptr;  // ping-pong scratch of the sum tree
    size_t part_cap = 0;        // doubles per scratch buffer
    double *sums = nullptr;     // [K][2*dim] sums of this shard
    double *coef = nullptr;     // [K][dim][2]
    size_t k_cap = 0;
    std::string error;
};

namespace {

thread_local std::string g_norm_create_error;

int nfail(mxv_norm *nm, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (nm)
        nm->error = buf;
    else
        g_norm_create_error = buf;
    return code;
}

#define NRM_HIP(nm, expr)                                                                               \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return nfail((nm), MXV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

bool dim_supported(int d) { return d == 1 || d == 2 || d == 3 || d == 4 || d == 6; }

int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

int ensure_capacity(mxv_norm *nm, int K, int64_t leaves, int V) {
    const size_t need = (size_t)K * (size_t)leaves * (size_t)V;
    if (need > nm->part_cap) {
        NRM_HIP(nm, hipStreamSynchronize(nm->stream));
        if (nm->part_a) NRM_HIP(nm, hipFree(nm->part_a));
        if (nm->part_b) NRM_HIP(nm, hipFree(nm->part_b));
        nm->part_a = nm->part_b = nullptr;
        NRM_HIP(nm, hipMalloc((void **)&nm->part_a, need * sizeof(double)));
        NRM_HIP(nm, hipMalloc((void **)&nm->part_b, need * sizeof(double)));
        nm->part_cap = need;
    }
    if ((size_t)K > nm->k_cap) {
        NRM_HIP(nm, hipStreamSynchronize(nm->stream));
        if (nm->sums) NRM_HIP(nm, hipFree(nm->sums));
        if (nm->coef) NRM_HIP(nm, hipFree(nm->coef));
        nm->sums = nm->coef = nullptr;
        NRM_HIP(nm, hipMalloc((void **)&nm->sums, (size_t)K * 2 * nm->dim * sizeof(double)));
        NRM_HIP(nm, hipMalloc((void **)&nm->coef, (size_t)K * 2 * nm->dim * sizeof(double)));
        nm->k_cap = (size_t)K;
    }
    return MXV_OK;
}

// folds partials [K][leaves][V] (in part_a, or the caller's `from`, which is only read) down to [K][V] written to dst (device)
int run_tree(mxv_norm *nm, int K, int64_t leaves, int V, double *dst, const double *from = nullptr) {
    const double *in = from ? from : nm->part_a;
    double *out = from ? nm->part_a : nm->part_b;
    while (true) {
        const int64_t groups = ceil_div(leaves, kTreeFan);
        double *target = groups == 1 ? dst : out;
        hipLaunchKernelGGL(tree_kernel, dim3((unsigned)groups, (unsigned)K), dim3(kThreads), 0, nm->stream, in, target, leaves, V);
        NRM_HIP(nm, hipGetLastError());
        if (groups == 1) break;
        leaves = groups;
        in = out;
        out = out == nm->part_a ? nm->part_b : nm->part_a;
    }
    return MXV_OK;
}

template <int O>
int launch_obs_sums(mxv_norm *nm, int K, const float *x, int64_t leaves) {
    hipLaunchKernelGGL(obs_sums_kernel<O>, dim3((unsigned)leaves, (unsigned)K), dim3(kThreads), 0, nm->stream, x, nm->n,
                       leaves, nm->part_a);
    NRM_HIP(nm, hipGetLastError());
    return MXV_OK;
}

template <int O>
int launch_obs_apply(mxv_norm *nm, int K, const float *x, void *y, int out_f32) {
    const dim3 grid((unsigned)ceil_div(nm->n, kThreads), (unsigned)K);
    if (out_f32)
        hipLaunchKernelGGL((obs_apply_kernel<O, float>), grid, dim3(kThreads), 0, nm->stream, x, (float *)y, nm->coef, nm->n);
    else
        hipLaunchKernelGGL((obs_apply_kernel<O, double>), grid, dim3(kThreads), 0, nm->stream, x, (double *)y, nm->coef, nm->n);
    NRM_HIP(nm, hipGetLastError());
    return MXV_OK;
}

#define DISPATCH_DIM(nm, fn, ...)                                                   \
    switch ((nm)->dim) {                                                            \
        case 1: return fn<1>(__VA_ARGS__);                                          \
        case 2: return fn<2>(__VA_ARGS__);                                          \
        case 3: return fn<3>(__VA_ARGS__);                                          \
        case 4: return fn<4>(__VA_ARGS__);                                          \
        case 6: return fn<6>(__VA_ARGS__);                                          \
        default: return nfail((nm), MXV_ERR_UNSUPPORTED, "dim %d", (nm)->dim);      \
    }

int dispatch_obs_sums(mxv_norm *nm, int K, const float *x, int64_t leaves) { DISPATCH_DIM(nm, launch_obs_sums, nm, K, x, leaves) }
int dispatch_obs_apply(mxv_norm *nm, int K, const float *x, void *y, int out_f32) { DISPATCH_DIM(nm, launch_obs_apply, nm, K, x, y, out_f32) }

int checks(mxv_norm *nm, int K) {
    if (!nm) return nfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_norm");
    if (K <= 0) return nfail(nm, MXV_ERR_INVALID_ARG, "K must be positive");
    if (K > 65535) return nfail(nm, MXV_ERR_INVALID_ARG, "K must be <= 65535 batches per call (grid.y)");
    NRM_HIP(nm, hipSetDevice(nm->device));
    return MXV_OK;
}

// caller-owned tensors on their element's natural boundary (observation rows: the vector width the kernels load them with)
int norm_aligned(mxv_norm *nm, const void *p, size_t bytes, const char *what) {
    if (p && ((uintptr_t)p & (bytes - 1)) != 0) return nfail(nm, MXV_ERR_INVALID_ARG, "%s pointer %p is not %zu-byte aligned", what, p, bytes);
    return MXV_OK;
}
size_t row_align(int dim, size_t elem) { return dim % 4 == 0 ? 4 * elem > 16 ? 16 : 4 * elem : (dim % 2 == 0 ? 2 * elem : elem); }

int run_scan(mxv_norm *nm, int K, const double *all_sums, int world, int64_t total_rows, double epsilon, int obs) {
    if (world < 1 || world > kMaxWorld) return nfail(nm, MXV_ERR_INVALID_ARG, "world must be in [1, %d]", kMaxWorld);
    if (total_rows <= 0) return nfail(nm, MXV_ERR_INVALID_ARG, "total_rows must be positive");
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kThreads), 0, nm->stream, all_sums, world, K, nm->dim, (double)total_rows, epsilon,
                       obs, nm->stat, nm->coef);
    NRM_HIP(nm, hipGetLastError());
    return MXV_OK;
}

}  // namespace

extern "C" {

int mxv_norm_create(int32_t device, int32_t dim, int64_t num_envs, void *stream, mxv_norm **out) {
    if (!out) return nfail(nullptr, MXV_ERR_INVALID_ARG, "NULL output pointer");
    *out = nullptr;
    if (!dim_supported(dim)) return nfail(nullptr, MXV_ERR_UNSUPPORTED, "dim must be one of 1, 2, 3, 4, 6 (got %d)", dim);
    if (num_envs <= 0) return nfail(nullptr, MXV_ERR_INVALID_ARG, "num_envs must be positive");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return nfail(nullptr, MXV_ERR_HIP, "no HIP device available (%s): the engine has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return nfail(nullptr, MXV_ERR_INVALID_ARG, "device %d out of range", device);
    mxv_norm *nm = new (std::nothrow) mxv_norm();
    if (!nm) return nfail(nullptr, MXV_ERR_INVALID_ARG, "out of host memory");
    nm->device = device;
    nm->dim = dim;
    nm->n = num_envs;
    nm->stream = (hipStream_t)stream;
    std::vector<double> init(2 * dim + 1, 0.0);
    for (int j = 0; j < dim; ++j) init[dim + j] = 1.0;  // RunningMeanStd.__init__: mean 0, var 1,
    init[2 * dim] = 1e-4;                                // count = epsilon = 1e-4 (normalize.py:12-15)
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess) err = hipMalloc((void **)&nm->stat, init.size() * sizeof(double));
    if (err == hipSuccess) err = hipMalloc((void **)&nm->returns, (size_t)num_envs * sizeof(double));
    if (err == hipSuccess) err = hipMemcpy(nm->stat, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemset(nm->returns, 0, (size_t)num_envs * sizeof(double));  // np.zeros(num_envs), :123
    if (err != hipSuccess) {
        nfail(nullptr, MXV_ERR_HIP, "mxv_norm_create: %s", hipGetErrorString(err));
        mxv_norm_destroy(nm);
        return MXV_ERR_HIP;
    }
    *out = nm;
    return MXV_OK;
}

int mxv_norm_destroy(mxv_norm *nm) {
    if (!nm) return MXV_OK;
    (void)hipSetDevice(nm->device);
    (void)hipStreamSynchronize(nm->stream);
    void *bufs[] = {nm->stat, nm->returns, nm->part_a, nm->part_b, nm->sums, nm->coef};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    delete nm;
    return MXV_OK;
}

const char *mxv_norm_last_error(const mxv_norm *nm) { return nm ? nm->error.c_str() : g_norm_create_error.c_str(); }

int mxv_norm_set_stream(mxv_norm *nm, void *stream) {
    if (!nm) return nfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_norm");
    NRM_HIP(nm, hipSetDevice(nm->device));
    NRM_HIP(nm, hipStreamSynchronize(nm->stream));
    nm->stream = (hipStream_t)stream;
    return MXV_OK;
}

int mxv_norm_get_state(mxv_norm *nm, double *mean_host, double *var_host, double *count_host, double *returns_host) {
    if (!nm) return nfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_norm");
    NRM_HIP(nm, hipSetDevice(nm->device));
    std::vector<double> st(2 * nm->dim + 1);
    NRM_HIP(nm, hipMemcpyAsync(st.data(), nm->stat, st.size() * sizeof(double), hipMemcpyDeviceToHost, nm->stream));
    if (returns_host)
        NRM_HIP(nm, hipMemcpyAsync(returns_host, nm->returns, (size_t)nm->n * sizeof(double), hipMemcpyDeviceToHost, nm->stream));
    NRM_HIP(nm, hipStreamSynchronize(nm->stream));
    for (int j = 0; j < nm->dim; ++j) {
        if (mean_host) mean_host[j] = st[j];
        if (var_host) var_host[j] = st[nm->dim + j];
    }
    if (count_host) *count_host = st[2 * nm->dim];
    return MXV_OK;
}

int mxv_norm_set_state(mxv_norm *nm, const double *mean_host, const double *var_host, double count, const double *returns_host) {
    if (!nm) return nfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_norm");
    if (!mean_host || !var_host) return nfail(nm, MXV_ERR_INVALID_ARG, "mean/var pointer is NULL");
    NRM_HIP(nm, hipSetDevice(nm->device));
    std::vector<double> st(2 * nm->dim + 1);
    for (int j = 0; j < nm->dim; ++j) {
        st[j] = mean_host[j];
        st[nm->dim + j] = var_host[j];
    }
    st[2 * nm->dim] = count;
    NRM_HIP(nm, hipMemcpyAsync(nm->stat, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice, nm->stream));
    if (returns_host)
        NRM_HIP(nm, hipMemcpyAsync(nm->returns, returns_host, (size_t)nm->n * sizeof(double), hipMemcpyHostToDevice, nm->stream));
    NRM_HIP(nm, hipStreamSynchronize(nm->stream));
    return MXV_OK;
}

int mxv_norm_obs_sums(mxv_norm *nm, int32_t K, const float *x_dev, double *sums_dev) {
    if (int rc = checks(nm, K)) return rc;
    if (!x_dev || !sums_dev) return nfail(nm, MXV_ERR_INVALID_ARG, "x/sums pointer is NULL");
    if (int rc = norm_aligned(nm, x_dev, row_align(nm->dim, 4), "x")) return rc;
    if (int rc = norm_aligned(nm, sums_dev, 8, "sums")) return rc;
    const int64_t leaves = ceil_div(nm->n, kObsLeafRows);
    if (int rc = ensure_capacity(nm, K, leaves, 2 * nm->dim)) return rc;
    if (int rc = dispatch_obs_sums(nm, K, x_dev, leaves)) return rc;
    return run_tree(nm, K, leaves, 2 * nm->dim, sums_dev);
}

int mxv_norm_obs_sums_partials(mxv_norm *nm, int32_t K, const double *partials_dev, int64_t leaves, double *sums_dev) {
    if (int rc = checks(nm, K)) return rc;
    if (!partials_dev || !sums_dev) return nfail(nm, MXV_ERR_INVALID_ARG, "partials/sums pointer is NULL");
    if (leaves < 1) return nfail(nm, MXV_ERR_INVALID_ARG, "leaves must be positive");
    if (int rc = norm_aligned(nm, partials_dev, 8, "partials")) return rc;
    if (int rc = norm_aligned(nm, sums_dev, 8, "sums")) return rc;
    if (int rc = ensure_capacity(nm, K, ceil_div(leaves, kTreeFan), 2 * nm->dim)) return rc;
    return run_tree(nm, K, leaves, 2 * nm->dim, sums_dev, partials_dev);
}

int mxv_norm_reward_sums_partials(mxv_norm *nm, int32_t K, const double *partials_dev, int64_t leaves, double *sums_dev) {
    if (int rc = checks(nm, K)) return rc;
    if (nm->dim != 1) return nfail(nm, MXV_ERR_INVALID_ARG, "reward sums need a 1-column mxv_norm");
    if (!partials_dev || !sums_dev) return nfail(nm, MXV_ERR_INVALID_ARG, "partials/sums pointer is NULL");
    if (leaves < 1) return nfail(nm, MXV_ERR_INVALID_ARG, "leaves must be positive");
    if (int rc = norm_aligned(nm, partials_dev, 8, "partials")) return rc;
    if (int rc = norm_aligned(nm, sums_dev, 8, "sums")) return rc;
    if (int rc = ensure_capacity(nm, K, ceil_div(leaves, kTreeFan), 2)) return rc;
    return run_tree(nm, K, leaves, 2, sums_dev, partials_dev);
}

int mxv_norm_returns_ptr(mxv_norm *nm, double **returns_dev) {
    if (!nm || !returns_dev) return nfail(nm, MXV_ERR_INVALID_ARG, "NULL mxv_norm or output pointer");
    *returns_dev = nm->returns;
    return MXV_OK;
}

int mxv_norm_obs_apply(mxv_norm *nm, int32_t K, const float *x_dev, void *y_dev, int32_t out_f32, double epsilon,
                       const double *all_sums_dev, int32_t world, int64_t total_rows) {
    if (int rc = checks(nm, K)) return rc;
    if (!x_dev || !y_dev || !all_sums_dev) return nfail(nm, MXV_ERR_INVALID_ARG, "x/y/sums pointer is NULL");
    if (int rc = norm_aligned(nm, x_dev, row_align(nm->dim, 4), "x")) return rc;
    if (int rc = norm_aligned(nm, y_dev, row_align(nm->dim, out_f32 ? 4 : 8), "y")) return rc;
    if (int rc = norm_aligned(nm, all_sums_dev, 8, "sums")) return rc;
    if (int rc = ensure_capacity(nm, K, 1, 2 * nm->dim)) return rc;
    if (int rc = run_scan(nm, K, all_sums_dev, world, total_rows, epsilon, 1)) return rc;
    return dispatch_obs_apply(nm, K, x_dev, y_dev, out_f32);
}

int mxv_norm_observations(mxv_norm *nm, int32_t K, const float *x_dev, void *y_dev, int32_t out_f32, double epsilon) {
    if (int rc = checks(nm, K)) return rc;
    if (!x_dev || !y_dev) return nfail(nm, MXV_ERR_INVALID_ARG, "x/y pointer is NULL");
    if (int rc = norm_aligned(nm, x_dev, row_align(nm->dim, 4), "x")) return rc;
    if (int rc = norm_aligned(nm, y_dev, row_align(nm->dim, out_f32 ? 4 : 8), "y")) return rc;
    const int64_t leaves = ceil_div(nm->n, kObsLeafRows);
    if (int rc = ensure_capacity(nm, K, leaves, 2 * nm->dim)) return rc;
    if (int rc = mxv_norm_obs_sums(nm, K, x_dev, nm->sums)) return rc;
    return mxv_norm_obs_apply(nm, K, x_dev, y_dev, out_f32, epsilon, nm->sums, 1, nm->n);
}

int mxv_norm_reward_sums(mxv_norm *nm, int32_t K, const void *reward_dev, int32_t reward_f32, const uint8_t *terminated_dev,
                         const uint8_t *truncated_dev, double gamma, double *sums_dev) {
    if (int rc = checks(nm, K)) return rc;
    if (nm->dim != 1) return nfail(nm, MXV_ERR_INVALID_ARG, "reward statistics need dim == 1 (RunningMeanStd(shape=()))");
    if (!reward_dev || !terminated_dev || !truncated_dev || !sums_dev)
        return nfail(nm, MXV_ERR_INVALID_ARG, "reward/terminated/truncated/sums pointer is NULL");
    if (int rc = norm_aligned(nm, reward_dev, reward_f32 ? 4 : 8, "reward")) return rc;
    if (int rc = norm_aligned(nm, sums_dev, 8, "sums")) return rc;
    const int64_t leaves = ceil_div(nm->n, kRewLeafEnvs);
    if (int rc = ensure_capacity(nm, K, leaves, 2)) return rc;
    // the vector loads (16 / 32 bytes of rewards, 4 of each flag array per lane) need tensors that start on those boundaries — true of
    // whole allocations, not of every view a caller may pass
    const int aligned = (uintptr_t)reward_dev % 32 == 0 && (uintptr_t)terminated_dev % 4 == 0 && (uintptr_t)truncated_dev % 4 == 0;
    if (reward_f32)
        hipLaunchKernelGGL(returns_sums_kernel<float>, dim3((unsigned)leaves), dim3(64), 0, nm->stream, (const float *)reward_dev,
                           terminated_dev, truncated_dev, nm->returns, nm->n, (int)K, gamma, leaves, nm->part_a, aligned);
    else
        hipLaunchKernelGGL(returns_sums_kernel<double>, dim3((unsigned)leaves), dim3(64), 0, nm->stream, (const double *)reward_dev,
                           terminated_dev, truncated_dev, nm->returns, nm->n, (int)K, gamma, leaves, nm->part_a, aligned);
    NRM_HIP(nm, hipGetLastError());
    return run_tree(nm, K, leaves, 2, sums_dev);
}

int mxv_norm_reward_apply(mxv_norm *nm, int32_t K, const void *reward_dev, int32_t reward_f32, void *out_dev, double epsilon,
                          const double *all_sums_dev, int32_t world, int64_t total_rows) {
    if (int rc = checks(nm, K)) return rc;
    if (nm->dim != 1) return nfail(nm, MXV_ERR_INVALID_ARG, "reward statistics need dim == 1 (RunningMeanStd(shape=()))");
    if (!reward_dev || !out_dev || !all_sums_dev) return nfail(nm, MXV_ERR_INVALID_ARG, "reward/out/sums pointer is NULL");
    if (int rc = norm_aligned(nm, reward_dev, reward_f32 ? 4 : 8, "reward")) return rc;
    if (int rc = norm_aligned(nm, out_dev, reward_f32 ? 4 : 8, "out")) return rc;
    if (int rc = norm_aligned(nm, all_sums_dev, 8, "sums")) return rc;
    if (int rc = ensure_capacity(nm, K, 1, 2)) return rc;
    if (int rc = run_scan(nm, K, all_sums_dev, world, total_rows, epsilon, 0)) return rc;
    const int V = reward_f32 ? 4 : 2;
    const bool vec = nm->n % V == 0 && ((uintptr_t)reward_dev | (uintptr_t)out_dev) % 16 == 0;
    const dim3 grid((unsigned)ceil_div(nm->n, (int64_t)kThreads * (vec ? V : 1)), (unsigned)K);
    if (reward_f32) {
        if (vec)
            hipLaunchKernelGGL((reward_apply_kernel<float, 4>), grid, dim3(kThreads), 0, nm->stream, (const float *)reward_dev,
                               (float *)out_dev, nm->coef, nm->n);
        else
            hipLaunchKernelGGL((reward_apply_kernel<float, 1>), grid, dim3(kThreads), 0, nm->stream, (const float *)reward_dev,
                               (float *)out_dev, nm->coef, nm->n);
    } else {
        if (vec)
            hipLaunchKernelGGL((reward_apply_kernel<double, 2>), grid, dim3(kThreads), 0, nm->stream, (const double *)reward_dev,
                               (double *)out_dev, nm->coef, nm->n);
        else
            hipLaunchKernelGGL((reward_apply_kernel<double, 1>), grid, dim3(kThreads), 0, nm->stream, (const double *)reward_dev,
                               (double *)out_dev, nm->coef, nm->n);
    }
    NRM_HIP(nm, hipGetLastError());
    return MXV_OK;
}

int mxv_norm_rewards(mxv_norm *nm, int32_t K, const void *reward_dev, int32_t reward_f32, const uint8_t *terminated_dev,
                     const uint8_t *truncated_dev, void *out_dev, double gamma, double epsilon) {
    if (int rc = checks(nm, K)) return rc;
    if (!out_dev) return nfail(nm, MXV_ERR_INVALID_ARG, "out pointer is NULL");
    if (int rc = norm_aligned(nm, out_dev, reward_f32 ? 4 : 8, "out")) return rc;
    const int64_t leaves = ceil_div(nm->n, kRewLeafEnvs);
    if (int rc = ensure_capacity(nm, K, leaves, 2)) return rc;
    if (int rc = mxv_norm_reward_sums(nm, K, reward_dev, reward_f32, terminated_dev, truncated_dev, gamma, nm->sums)) return rc;
    return mxv_norm_reward_apply(nm, K, reward_dev, reward_f32, out_dev, epsilon, nm->sums, 1, nm->n);
}

}  // extern "C"
