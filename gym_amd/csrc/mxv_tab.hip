// mxv_tab.hip — SURVEY.md §8(f)-4: the reference's tabular toy_text environments (FrozenLake-v1, FrozenLake8x8-v1,
// Taxi-v3, CliffWalking-v0) as ONE table-driven gfx950 engine behind the mxv_tab_* C ABI (include/mxv.h).
//
// All three reference classes share one step() (gym/envs/toy_text/frozen_lake.py:247-256, taxi.py:254-263,
// cliffwalking.py:148-157):   transitions = P[s][a];  i = categorical_sample([t[0] for t in transitions], np_random)
//                             p, s, r, t = transitions[i];  return int(s), r, t, False, {"prob": p}
// and one reset() (frozen_lake.py:258-270, taxi.py:265-278, cliffwalking.py:159-166):
//                             s = categorical_sample(initial_state_distrib, np_random)
// with categorical_sample = argmax(cumsum(prob_n) > np_random.random()) (toy_text/utils.py:4-8).  The MDP itself is the
// table P, which the HOST builds (gym_amd/toy_text.py restates the three __init__ constructions) and hands over as
// dense arrays [S][A][M] (M = the longest transition list; shorter lists are padded with cum_prob = -1, which never
// compares greater than a uniform in [0,1), so "first index whose cumulative probability exceeds u, 0 if none" is
// preserved).  TimeLimit (gym/wrappers/time_limit.py:39-68) and SyncVectorEnv's autoreset + final_observation
// (gym/vector/sync_vector_env.py:135-169) are fused exactly as in the classic-control kernels.
//
// Kernel: one env per lane, the whole table staged in LDS once per workgroup (FrozenLake8x8: 24 KiB, Taxi: 40 KiB);
// per env-step: one Philox4x32-10 call for the env's transition/reset uniforms, one (shared by 4 envs) for the action,
// <= M + 3 LDS reads, and the outputs — obs int64, reward f64, 2 flag bytes, prob f64, action int64: 34 B — which is
// what bounds it (HBM).  K steps per launch keep state + elapsed in registers (mxv_tab_rollout).
//
// RNG contract (Philox4x32-10, counter based): actions as in the classic engine (key = action_seed, ctr = (g, t, stream 1));
// transition stream: key = per-env seed, ctr = (b_lo, b_hi, 0, 3 << 28) with b = t >> 1: words (x, y) serve the even step
// 2b, (z, w) the odd step 2b+1 — first word: the step's uniform, second: the uniform of an autoreset inside that step (one
// Philox call per env per TWO steps); explicit reset: the classic reset stream (ctr = (t_lo, t_hi, r, 2 << 28)),
// word x.  uniform = (word + 0.5) * 2^-32.  A caller may instead inject the uniforms (mxv_tab_step, `uniforms_dev`):
// that is how the parity tests replay the reference's own PCG64 draws bit for bit.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mxv_device.hpp"

using namespace mxv;

namespace {

constexpr int kTabBlock = 256;
constexpr uint32_t kStreamTransition = 3u;
constexpr unsigned kTabXcds = 8;

struct TabArgs {
    int32_t *state;          // [N] current state index
    int32_t *elapsed;        // [N] TimeLimit counters
    const uint64_t *seeds;   // per-env seeds or nullptr
    // the MDP (global copies; staged into LDS when they fit)
    const double *cum;       // [S*A*M] cumulative transition probabilities (padding = -1) or nullptr when M == 1
    const double *prob;      // [S*A*M] transition probabilities or nullptr when every probability is 1.0
    const double *reward;    // [S*A*M]
    const int32_t *nt;       // [S*A*M] next_state | terminated << 31
    const double *init_cum;  // [S] cumulative initial-state distribution
    int32_t S, A, M, log2S;
    int32_t single_start;    // >= 0: the initial distribution is a point mass on this state (FrozenLake, CliffWalking): no search
    // step I/O
    // integer tensors are int64 and real ones float64 — int32 / float32 in a COMPACT launch (MXV_TAB_FLAG_COMPACT, trajectory calls)
    const void *actions;     // [N] (or [K][N] tape) or nullptr -> Philox action stream
    void *actions_out;       // optional
    const double *uniforms;  // optional [2][N]: (transition uniform, autoreset uniform) per env; nullptr -> Philox
    void *obs;               // [N] / [K][N]
    void *reward_out;        // may be nullptr
    uint8_t *terminated, *truncated;  // may be nullptr
    void *prob_out;          // may be nullptr: info["prob"] (1.0 for envs that were reset in this step)
    void *final_obs;         // may be nullptr: terminal state of finished envs (untouched elsewhere)
    void *final_prob;        // may be nullptr: info["final_info"]["prob"] of finished envs
    int32_t *err;
    int64_t n;
    uint64_t env0, base_seed, action_seed, t;
    const uint64_t *t_dev;   // device clock (mxv_tab_set_device_clock): the step index = t + *t_dev; nullptr: t
    int32_t max_steps, K;
    int64_t slice;           // 0 or N (per-step trajectory outputs)
    int64_t act_slice;       // 0 or N (action tape)
    // episode statistics (gym/wrappers/record_episode_statistics.py:96-151), all nullptr when disabled (mxv_tab_episode_stats)
    float *ep_acc;           // [N] running episode return, float32 like the reference's accumulator
    float *ep_return_out;    // [N] / [K][N]: episode return, written only where terminated | truncated
    int32_t *ep_length_out;  // [N] / [K][N]: episode length (= the TimeLimit counter), written only where terminated | truncated
};

__device__ __forceinline__ unsigned tab_tile(unsigned bid, unsigned ntiles) {  // XCD x owns the x-th contiguous eighth
    const unsigned x = bid % kTabXcds, idx = bid / kTabXcds;
    const unsigned base = ntiles / kTabXcds, rem = ntiles % kTabXcds;
    return x * base + (x < rem ? x : rem) + idx;
}

__device__ __forceinline__ U4 transition_words(uint64_t seed, uint64_t b) {  // b = t >> 1
    U4 c;
    c.x = (uint32_t)b;
    c.y = (uint32_t)(b >> 32);
    c.z = 0;
    c.w = (kStreamTransition << 28);
    return philox4x32_10_vkey(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// categorical_sample(initial_state_distrib): first index whose cumulative probability exceeds u, 0 if none.
__device__ __forceinline__ int32_t sample_initial(const double *init_cum, int32_t S, int32_t log2S, double u) {
    int32_t lo = 0, hi = S;
    for (int it = 0; it <= log2S; ++it) {
        if (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (init_cum[mid] > u) hi = mid;
            else lo = mid + 1;
        }
    }
    return lo < S ? lo : 0;
}

template <bool LDS_TABLE, bool COMPACT>
__global__ void __launch_bounds__(kTabBlock) tab_step_kernel(TabArgs a) {
    using IT = std::conditional_t<COMPACT, int32_t, int64_t>;   // element type of the integer step tensors
    using RT = std::conditional_t<COMPACT, float, double>;      // ... of the real ones
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int entries = a.S * a.A * a.M;
    const double *cum = a.cum, *prob = a.prob, *reward = a.reward, *init_cum = a.init_cum;
    const int32_t *nt = a.nt;
    if (LDS_TABLE) {
        double *p = smem;
        double *l_cum = nullptr, *l_prob = nullptr;
        if (a.cum) { l_cum = p; p += entries; }
        if (a.prob) { l_prob = p; p += entries; }
        double *l_rew = p; p += entries;
        double *l_init = p; p += a.S;
        int32_t *l_nt = reinterpret_cast<int32_t *>(p);
        for (int i = tid; i < entries; i += kTabBlock) {
            if (l_cum) l_cum[i] = a.cum[i];
            if (l_prob) l_prob[i] = a.prob[i];
            l_rew[i] = a.reward[i];
            l_nt[i] = a.nt[i];
        }
        for (int i = tid; i < a.S; i += kTabBlock) l_init[i] = a.init_cum[i];
        __syncthreads();
        cum = l_cum; prob = l_prob; reward = l_rew; init_cum = l_init; nt = l_nt;
    }
    const int64_t e = (int64_t)tab_tile(blockIdx.x, gridDim.x) * kTabBlock + tid;
    const bool valid = e < a.n;  // lanes past the end stay in the loop: their quad partners need their action words
    const uint64_t ge = a.env0 + (uint64_t)e;
    const uint64_t seed = (a.seeds && valid) ? a.seeds[e] : a.base_seed + ge;
    int32_t s = valid ? a.state[e] : 0, el = valid ? a.elapsed[e] : 0;
    float er = (a.ep_acc && valid) ? a.ep_acc[e] : 0.0f;   // RecordEpisodeStatistics.episode_returns of this env
    // Philox caches.  Actions: one call yields the words of the 4 envs of group g = env >> 2 at ONE step, so the four
    // lanes of a quad (= one group) each evaluate a different step of the aligned block 4*(t >> 2) .. +3 and trade words
    // through a 4 x 4 transpose inside the quad (quad_transpose: two DPP butterfly stages): one call per lane per four steps.  Transitions: one call per two steps (see the contract).
    const uint32_t q = (uint32_t)(ge & 3);
    uint64_t act_block = ~0ull, tr_block = ~0ull;
    uint32_t act_word[4] = {0, 0, 0, 0};
    U4 tw{0, 0, 0, 0};
    const uint64_t t_base = a.t + (a.t_dev ? *a.t_dev : 0);
    mxv::settle_entry_loads();
    for (int k = 0; k < a.K; ++k) {
        const uint64_t t = t_base + (uint64_t)k;
        const int64_t o = (int64_t)k * a.slice + e;
        // ---- action: caller's, or Discrete(A).sample() from the Philox action stream ----
        int64_t act;
        if (a.actions) {
            if (!valid) continue;
            act = (int64_t) static_cast<const IT *>(a.actions)[(int64_t)k * a.act_slice + e];
            if (act < 0 || act >= a.A) {  // `assert self.action_space.contains(a)`-class error: latch, leave the env unstepped
                *reinterpret_cast<volatile int32_t *>(a.err) = 1;  // single-bit code: a plain store (the word may live in pinned host memory)
                continue;
            }
        } else {
            if ((t >> 2) != act_block) {  // uniform across the launch: every lane refills its cache at the same step
                act_block = t >> 2;
                const U4 w = action_words(a.action_seed, (act_block << 2) + q, ge >> 2);
                act_word[0] = w.x; act_word[1] = w.y; act_word[2] = w.z; act_word[3] = w.w;
                quad_transpose(act_word, q);   // lane q evaluated step q of the block for the quad's four envs -> its own env's words for steps 0..3
            }
            const uint32_t j = (uint32_t)(t & 3);
            const uint32_t word = j == 0 ? act_word[0] : (j == 1 ? act_word[1] : (j == 2 ? act_word[2] : act_word[3]));
            act = (int64_t)(((uint64_t)word * (uint64_t)a.A) >> 32);
            if (!valid) continue;
            if (a.actions_out) static_cast<IT *>(a.actions_out)[o] = (IT)act;
        }
        // ---- the step's uniforms ----
        double u_step, u_reset;
        if (a.uniforms) {
            u_step = a.uniforms[e];
            u_reset = a.uniforms[a.n + e];
        } else {
            if ((t >> 1) != tr_block) {
                tr_block = t >> 1;
                tw = transition_words(seed, tr_block);
            }
            u_step = u01((t & 1) ? tw.z : tw.x);
            u_reset = u01((t & 1) ? tw.w : tw.y);
        }
        // ---- categorical_sample over P[s][act] (toy_text/utils.py:4-8) ----
        const int base = (s * a.A + (int)act) * a.M;
        int idx = 0;
        if (cum) {
            bool found = false;
            for (int i = 0; i < a.M; ++i) {
                const bool hit = !found && cum[base + i] > u_step;
                idx = hit ? i : idx;
                found = found || hit;
            }
        }
        const int j = base + idx;
        const int32_t packed = nt[j];
        const int32_t ns = packed & 0x7fffffff;
        const bool term = packed < 0;
        double p = prob ? prob[j] : 1.0;
        const double rew = reward[j];
        el += 1;                                                   // TimeLimit.step, time_limit.py:50-53
        const bool trunc = a.max_steps > 0 && el >= a.max_steps;
        s = ns;
        if (a.ep_acc) {  // record_episode_statistics.py:119-143: float32 array += float64 reward; lengths are the TimeLimit counter
            er = (float)((double)er + rew);
            const bool fin = term || trunc;
            if (__any(fin)) {   // a wave with a finished episode stores its whole row segment (whole lines, zeros elsewhere): see tab_traj_kernel
                if (a.ep_return_out) a.ep_return_out[o] = fin ? er : 0.0f;
                if (a.ep_length_out) a.ep_length_out[o] = fin ? el : 0;
            }
            er = fin ? 0.0f : er;
        }
        if (term || trunc) {  // sync_vector_env.py:152-156: the returned observation/info are the reset's
            if (a.final_obs) static_cast<IT *>(a.final_obs)[o] = (IT)ns;
            if (a.final_prob) static_cast<RT *>(a.final_prob)[o] = (RT)p;
            s = a.single_start >= 0 ? a.single_start : sample_initial(init_cum, a.S, a.log2S, u_reset);
            el = 0;
            p = 1.0;                                               // reset() returns {"prob": 1}
        }
        static_cast<IT *>(a.obs)[o] = (IT)s;
        if (a.reward_out) static_cast<RT *>(a.reward_out)[o] = (RT)rew;
        if (a.terminated) a.terminated[o] = term ? 1 : 0;
        if (a.truncated) a.truncated[o] = trunc ? 1 : 0;
        if (a.prob_out) static_cast<RT *>(a.prob_out)[o] = (RT)p;
    }
    if (valid) {
        a.state[e] = s;
        a.elapsed[e] = el;
        if (a.ep_acc) a.ep_acc[e] = er;
    }
}

// ---- the trajectory fast path: sampled actions, every per-step output present (what mxv_tab_rollout's [K][N] launches of the registered
// envs are) -------------------------------------------------------------------------------------------------------------------------
// Same streams and the same values as tab_step_kernel, bit for bit (tests/test_gpu_toytext.py holds the two against each other and
// against the CPU restatement); what differs is the work per env-step:
//   * categorical_sample in the integer domain.  cum > u with u = (w + 0.5) * 2^-32 (exact in fp64) <=> w < T, T = ceil(cum * 2^32 - 0.5)
//     (cum * 2^32 - 0.5 is exact too).  The host packs T - 1 per transition; the index of the first cumulative probability that exceeds
//     u is then the NUMBER of thresholds below w: M - 1 unsigned compares, no fp64 conversion, no loop.  Lists whose last cumulative
//     probability does not reach 1 - 2^-33, zero-probability heads, or rewards that are not float32 values keep the general kernel
//     (pack_fast_table decides at create time).
//   * one LDS read per table level: {thresholds} -> {next state | terminated, reward, prob} packed in 16 (M > 1) or 8 bytes (M == 1);
//   * M == 1 (Taxi, CliffWalking, non-slippery lakes): the transition uniform is never used, so the env's Philox call is made only in the
//     steps where some env of the wave needs its autoreset uniform — and never for a point-mass initial distribution;
//   * the four action words of a quad change lanes through two DPP butterfly stages (a 4 x 4 transpose) instead of three shuffles and
//     select chains; the step loop is unrolled over the aligned block of four steps those words serve;
//   * no pointer tests in the loop, stores with the block's scalar base + one 32-bit lane offset each.
struct TabTrajArgs {
    int32_t *state, *elapsed;
    const uint64_t *seeds;
    const uint32_t *table;   // packed by pack_fast_table
    int32_t table_words;     // multiple of 4
    int32_t A;
    int32_t ent_off;         // word offset of the transition entries
    int32_t init_off, S1, init_iters;  // initial distribution that is no point mass: thresholds [S1 + 1], search iterations
    int32_t start;           // point mass: the start state
    char *actions, *obs, *reward, *prob;  // [K][N]
    uint8_t *terminated, *truncated;
    int64_t n;
    uint64_t env0, base_seed, action_seed, t;
    const uint64_t *t_dev;
    int32_t max_steps, K;
    uint32_t tile_base;      // ALLV == false: the (one) ragged block's tile index
    // episode statistics (mxv_tab_episode_stats), all nullptr when disabled: a wave-uniform test per step, one float32 add where enabled
    float *ep_acc;           // [N]
    float *ep_return_out;    // [K][N], written only where terminated | truncated
    int32_t *ep_length_out;  // [K][N]
};

__device__ __forceinline__ uint32_t tab_pin32(uint32_t v) {  // see pin32 in mxv_kernels.hip: keeps the store's saddr + 32-bit voffset form
    asm volatile("" : "+v"(v));
    return v;
}

template <int M_T, bool COMPACT, bool SINGLE, bool ALLV>
__global__ void __launch_bounds__(kTabBlock) tab_traj_kernel(TabTrajArgs a) {
    static_assert(M_T == 1 || M_T == 3, "instantiated for the transition-list lengths of the registered envs");
    constexpr uint32_t IB = COMPACT ? 4u : 8u;  // bytes of an integer / a real element of the trajectory tensors
    extern __shared__ uint32_t tbl[];
    const uint32_t tid = threadIdx.x;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.table);
        uint4 *dst = reinterpret_cast<uint4 *>(tbl);
        for (int i = (int)tid; i < a.table_words / 4; i += kTabBlock) dst[i] = src[i];
        __syncthreads();
    }
    const unsigned tile = ALLV ? tab_tile(blockIdx.x, gridDim.x) : a.tile_base;
    const int64_t tile0 = (int64_t)tile * kTabBlock;
    const int64_t e = tile0 + tid;
    const bool valid = ALLV || e < a.n;  // lanes past the end keep stepping (their quad partners need their action words)
    const uint64_t ge = a.env0 + (uint64_t)e;
    uint64_t seed = a.base_seed + ge;
    int32_t s = 0, el = 0;
    if (valid) {
        if (a.seeds) seed = a.seeds[e];
        s = a.state[e];
        el = a.elapsed[e];
    }
    const uint32_t q = (uint32_t)(ge & 3);
    const bool stats = a.ep_acc != nullptr;
    float er = (stats && valid) ? a.ep_acc[e] : 0.0f;
    float *p_epr = a.ep_return_out ? a.ep_return_out + tile0 : nullptr;
    int32_t *p_epl = a.ep_length_out ? a.ep_length_out + tile0 : nullptr;
    const int32_t max_eff = a.max_steps > 0 ? a.max_steps : 0x7fffffff;
    char *p_act = a.actions + tile0 * IB, *p_obs = a.obs + tile0 * IB, *p_rew = a.reward + tile0 * IB, *p_prob = a.prob + tile0 * IB;
    uint8_t *p_term = a.terminated + tile0, *p_trunc = a.truncated + tile0;
    const int64_t slice_w = a.n * (int64_t)IB, slice_b = a.n;
    const uint32_t off_w = tid * IB, off_b = tid;
    const uint64_t t0 = a.t + (a.t_dev ? *a.t_dev : 0), t1 = t0 + (uint64_t)a.K;
    mxv::settle_entry_loads();
    for (uint64_t blk = t0 >> 2; blk <= ((t1 - 1) >> 2); ++blk) {
        uint32_t aw[4];
        {
            const U4 w = action_words(a.action_seed, (blk << 2) + q, ge >> 2);  // this lane: step q of the block, the quad's 4 envs
            aw[0] = w.x; aw[1] = w.y; aw[2] = w.z; aw[3] = w.w;
            quad_transpose(aw, q);                                                // -> this lane's env at steps 0..3 of the block
        }
        U4 tw{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t t = (blk << 2) + (uint64_t)j;
            if (M_T > 1 && (j & 1) == 0) tw = transition_words(seed, t >> 1);
            if (t < t0 || t >= t1) continue;  // wave-uniform: only the first and the last block of a launch can be partial
            const uint32_t act = __umulhi(aw[j], (uint32_t)a.A);
            const uint32_t sa = (uint32_t)s * (uint32_t)a.A + act;
            uint32_t nt, rew_bits, p_lo = 0, p_hi = 0;
            if constexpr (M_T == 1) {
                const uint2 en = *reinterpret_cast<const uint2 *>(&tbl[a.ent_off + sa * 2u]);
                nt = en.x; rew_bits = en.y;
            } else {
                const uint32_t w = (j & 1) ? tw.z : tw.x;
                const uint2 th = *reinterpret_cast<const uint2 *>(&tbl[sa * 2u]);
                const uint32_t idx = (w > th.x ? 1u : 0u) + (w > th.y ? 1u : 0u);
                const uint4 en = *reinterpret_cast<const uint4 *>(&tbl[a.ent_off + (sa * 3u + idx) * 4u]);
                nt = en.x; rew_bits = en.y; p_lo = en.z; p_hi = en.w;
            }
            el += 1;                                                   // TimeLimit.step, time_limit.py:50-53
            const bool trunc = el >= max_eff;
            const bool term = (int32_t)nt < 0;
            const bool done = term || trunc;
            int32_t ns = (int32_t)(nt & 0x7fffffffu);
            if constexpr (SINGLE) {
                ns = done ? a.start : ns;                              // sync_vector_env.py:152-156: the reset's observation
            } else {
                if (__any(done)) {
                    uint32_t wr;
                    if constexpr (M_T == 1) {
                        // the env's Philox call of this step pair, made HERE only (the asm keeps the compiler from speculating the ten
                        // rounds into the path every step takes)
                        const U4 x = transition_words(((uint64_t)tab_pin32((uint32_t)(seed >> 32)) << 32) | tab_pin32((uint32_t)seed), t >> 1);
                        wr = (j & 1) ? x.w : x.y;
                    } else {
                        wr = (j & 1) ? tw.w : tw.y;
                    }
                    int32_t lo = 0, hi = a.S1;                         // first state whose cumulative probability exceeds u
                    for (int it = 0; it < a.init_iters; ++it) {
                        const int32_t mid = (lo + hi) >> 1;
                        const bool live = lo < hi, hit = wr < tbl[a.init_off + mid];
                        hi = (live && hit) ? mid : hi;
                        lo = (live && !hit) ? mid + 1 : lo;
                    }
                    ns = done ? lo : ns;
                }
            }
            s = ns;
            if (stats) {   // record_episode_statistics.py:119-143 (the table's rewards are float32 values: pack_fast_table)
                er = (float)((double)er + (double)__uint_as_float(rew_bits));
                // Stores by the few lanes whose episode ended are partial-line writes (read-modify-write at the memory side): with 3 % of
                // the envs finishing per step (FrozenLake8x8) two such streams cost 3.4 us on a 5.6-us step.  A wave in which ANY episode
                // ended therefore stores its 64 entries — the value where it ended, zero elsewhere — as two whole lines; waves without one
                // store nothing (Taxi: 94 % of them).  Entries are MEANINGFUL only where terminated | truncated, as the header says.
                if (__any(done) && (ALLV || valid)) {
                    if (p_epr) p_epr[tid] = done ? er : 0.0f;
                    if (p_epl) p_epl[tid] = done ? el : 0;
                }
                er = done ? 0.0f : er;
                if (p_epr) p_epr += a.n;
                if (p_epl) p_epl += a.n;
            }
            el = done ? 0 : el;
            if (ALLV || valid) {
                const float rew = __uint_as_float(rew_bits);
                if constexpr (COMPACT) {
                    *reinterpret_cast<int32_t *>(p_act + tab_pin32(off_w)) = (int32_t)act;
                    *reinterpret_cast<float *>(p_rew + tab_pin32(off_w)) = rew;
                    *reinterpret_cast<int32_t *>(p_obs + tab_pin32(off_w)) = s;
                    float p = 1.0f;
                    if constexpr (M_T > 1) p = done ? 1.0f : __uint_as_float(p_lo);
                    *reinterpret_cast<float *>(p_prob + tab_pin32(off_w)) = p;
                } else {
                    *reinterpret_cast<int64_t *>(p_act + tab_pin32(off_w)) = (int64_t)act;
                    *reinterpret_cast<double *>(p_rew + tab_pin32(off_w)) = (double)rew;
                    *reinterpret_cast<int64_t *>(p_obs + tab_pin32(off_w)) = (int64_t)s;
                    double p = 1.0;
                    if constexpr (M_T > 1) p = done ? 1.0 : __hiloint2double((int)p_hi, (int)p_lo);
                    *reinterpret_cast<double *>(p_prob + tab_pin32(off_w)) = p;
                }
                p_term[tab_pin32(off_b)] = term ? 1 : 0;
                p_trunc[tab_pin32(off_b)] = trunc ? 1 : 0;
            }
            p_act += slice_w; p_obs += slice_w; p_rew += slice_w; p_prob += slice_w;
            p_term += slice_b; p_trunc += slice_b;
        }
    }
    if (valid) {
        a.state[e] = s;
        a.elapsed[e] = el;
        if (stats) a.ep_acc[e] = er;
    }
}

__global__ void tab_set_word_kernel(uint64_t *dst, uint64_t v) { *dst = v; }
__global__ void tab_add_word_kernel(uint64_t *dst, uint64_t d) { *dst += d; }

struct TabResetArgs {
    int32_t *state, *elapsed;
    const uint64_t *seeds;
    const uint8_t *mask;
    const double *init_cum;
    int64_t *obs;
    int32_t S, log2S;
    int64_t n;
    uint64_t env0, base_seed, t;
    const uint64_t *t_dev;
    uint32_t r;
    float *ep_acc;           // may be nullptr: running episode returns, zeroed for the envs being reset (record_episode_statistics.py:91-94)
};

__global__ void __launch_bounds__(kTabBlock) tab_reset_kernel(TabResetArgs a) {
    const int64_t e = (int64_t)blockIdx.x * kTabBlock + threadIdx.x;
    if (e >= a.n) return;
    if (a.mask && !a.mask[e]) {
        if (a.obs) a.obs[e] = (int64_t)a.state[e];
        return;
    }
    const uint64_t seed = a.seeds ? a.seeds[e] : a.base_seed + a.env0 + (uint64_t)e;
    const U4 w = reset_words(seed, a.t + (a.t_dev ? *a.t_dev : 0), a.r);
    const int32_t s = sample_initial(a.init_cum, a.S, a.log2S, u01(w.x));
    a.state[e] = s;
    a.elapsed[e] = 0;
    if (a.ep_acc) a.ep_acc[e] = 0.0f;
    if (a.obs) a.obs[e] = (int64_t)s;
}

}  // namespace

struct mxv_tab {
    mxv_tab_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int32_t *state = nullptr, *elapsed = nullptr, *err = nullptr, *nt = nullptr;
    uint64_t *seeds = nullptr;
    double *cum = nullptr, *prob = nullptr, *reward = nullptr, *init_cum = nullptr;
    size_t lds_bytes = 0;
    bool lds_table = false;
    // trajectory fast path (tab_traj_kernel): the packed table and where its parts start; fast_M == 0: not eligible
    uint32_t *fast_tbl = nullptr;
    int fast_M = 0, fast_words = 0, fast_ent_off = 0, fast_init_off = 0, fast_S1 = 0, fast_iters = 0;
    int last_kernel = MXV_TAB_KERNEL_NONE;
    int log2S = 0;
    int single_start = -1;
    uint64_t base_seed = 0, action_seed = 0, t = 0;
    uint64_t *t_dev = nullptr;   // device clock (mxv_tab_set_device_clock)
    bool dev_clock = false;
    uint32_t r = 0;
    bool was_reset = false;
    // episode statistics (mxv_tab_episode_stats): running returns, the caller's trajectory outputs, dense staging of host steps
    float *ep_acc = nullptr, *ep_return_out = nullptr, *st_ep_r = nullptr;
    int32_t *ep_length_out = nullptr, *st_ep_l = nullptr;
    bool ep_host_step = false;   // the launch in flight is a host step: its statistics go to the staging arrays
    // staging for *_host calls
    int64_t *st_actions = nullptr, *st_obs = nullptr, *st_final = nullptr;
    double *st_reward = nullptr, *st_prob = nullptr, *st_fprob = nullptr, *st_uniforms = nullptr;
    uint8_t *st_term = nullptr, *st_trunc = nullptr, *st_mask = nullptr;
    // small vector envs (all host-step I/O <= 2 MiB): the staging arrays are slices of ONE pinned, device-mapped host block the
    // kernel reads and writes over PCIe — a host step is one launch and one synchronisation instead of eight copies (the
    // latency-bound regime: 8 FrozenLake envs 148 -> ~35 us per step; same scheme as mxv_api.cpp's I/O block)
    char *hm_block = nullptr;
    int32_t *hm_err = nullptr;
    bool hostmap = false, err_in_block = false;
    std::string error;
};

namespace {

thread_local std::string g_tab_create_error;

int tfail(mxv_tab *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h)
        h->error = buf;
    else
        g_tab_create_error = buf;
    return code;
}

#define TAB_HIP(h, expr)                                                                             \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return tfail((h), MXV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define TAB_CHECK(h) \
    if (!(h)) return tfail(nullptr, MXV_ERR_INVALID_ARG, "NULL mxv_tab")

// see mxv_set_device_clock (include/mxv.h): the step index on the device, advanced on the stream
int tab_clock_add(mxv_tab *h, int64_t delta) {
    h->t += (uint64_t)delta;
    if (h->dev_clock) {
        hipLaunchKernelGGL(tab_add_word_kernel, dim3(1), dim3(1), 0, h->stream, h->t_dev, (uint64_t)delta);
        TAB_HIP(h, hipGetLastError());
    }
    return MXV_OK;
}
int tab_clock_set(mxv_tab *h) {
    if (h->dev_clock) {
        hipLaunchKernelGGL(tab_set_word_kernel, dim3(1), dim3(1), 0, h->stream, h->t_dev, h->t);
        TAB_HIP(h, hipGetLastError());
    }
    return MXV_OK;
}

int tab_check_latched(mxv_tab *h) {
    int32_t e = 0;
    TAB_HIP(h, hipMemcpyAsync(&e, h->err, sizeof e, hipMemcpyDeviceToHost, h->stream));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    if (e != 0) {
        TAB_HIP(h, hipMemsetAsync(h->err, 0, sizeof(int32_t), h->stream));
        return tfail(h, MXV_ERR_INVALID_ACTION, "discrete action outside [0, %d) (Discrete.contains)", h->cfg.num_actions);
    }
    return MXV_OK;
}

// caller-owned tensors on their element's natural boundary (see check_aligned in mxv_api.cpp): refused by name, not torn
int tab_aligned(mxv_tab *h, const void *p, size_t bytes, const char *what) {
    if (p && ((uintptr_t)p & (bytes - 1)) != 0) return tfail(h, MXV_ERR_INVALID_ARG, "%s pointer %p is not %zu-byte aligned", what, p, bytes);
    return MXV_OK;
}
int tab_check_buffers(mxv_tab *h, bool compact, const void *actions, const void *actions_out, const void *uniforms, const void *obs,
                      const void *reward, const void *prob, const void *final_obs, const void *final_prob) {
    const size_t w = compact ? 4 : 8;
    const struct { const void *p; size_t b; const char *what; } t[] = {
        {actions, w, "actions"}, {actions_out, w, "actions_out"}, {uniforms, 8, "uniforms"}, {obs, w, "obs"}, {reward, w, "reward"},
        {prob, w, "prob"}, {final_obs, w, "final_obs"}, {final_prob, w, "final_prob"}, {h->ep_return_out, 4, "episode return"},
        {h->ep_length_out, 4, "episode length"}};
    for (const auto &e : t)
        if (int rc = tab_aligned(h, e.p, e.b, e.what)) return rc;
    return MXV_OK;
}

int tab_launch(mxv_tab *h, int K, int64_t slice, const void *actions, int64_t act_slice, void *actions_out,
               const double *uniforms, void *obs, void *reward, uint8_t *term, uint8_t *trunc, void *prob,
               void *final_obs, void *final_prob, bool compact = false) {
    if (!h->was_reset)
        return tfail(h, MXV_ERR_RESET_NEEDED, "Cannot call step before calling reset (gym.error.ResetNeeded)");
    if (!obs) return tfail(h, MXV_ERR_INVALID_ARG, "obs pointer is NULL");
    if (K <= 0) return tfail(h, MXV_ERR_INVALID_ARG, "K must be positive");
    if (int rc = tab_check_buffers(h, compact, actions, actions_out, uniforms, obs, reward, prob, final_obs, final_prob)) return rc;
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    TabArgs a{};
    a.state = h->state; a.elapsed = h->elapsed; a.seeds = h->seeds;
    a.cum = h->cum; a.prob = h->prob; a.reward = h->reward; a.nt = h->nt; a.init_cum = h->init_cum;
    a.S = h->cfg.num_states; a.A = h->cfg.num_actions; a.M = h->cfg.max_transitions; a.log2S = h->log2S;
    a.single_start = h->single_start;
    a.actions = actions; a.actions_out = actions_out; a.uniforms = uniforms;
    a.obs = obs; a.reward_out = reward; a.terminated = term; a.truncated = trunc; a.prob_out = prob;
    a.final_obs = final_obs; a.final_prob = final_prob;
    a.err = h->err_in_block ? h->hm_err : h->err; a.n = h->cfg.num_envs; a.env0 = (uint64_t)h->cfg.env_offset;
    a.base_seed = h->base_seed; a.action_seed = h->action_seed; a.t = h->dev_clock ? 0 : h->t; a.t_dev = h->dev_clock ? h->t_dev : nullptr;
    a.max_steps = h->cfg.max_episode_steps; a.K = K; a.slice = slice; a.act_slice = act_slice;
    a.ep_acc = h->ep_acc;
    a.ep_return_out = h->ep_acc ? (h->ep_host_step ? h->st_ep_r : h->ep_return_out) : nullptr;
    a.ep_length_out = h->ep_acc ? (h->ep_host_step ? h->st_ep_l : h->ep_length_out) : nullptr;
    const unsigned blocks = (unsigned)((h->cfg.num_envs + kTabBlock - 1) / kTabBlock);
    if (h->lds_table) {
        if (compact)
            hipLaunchKernelGGL((tab_step_kernel<true, true>), dim3(blocks), dim3(kTabBlock), h->lds_bytes, h->stream, a);
        else
            hipLaunchKernelGGL((tab_step_kernel<true, false>), dim3(blocks), dim3(kTabBlock), h->lds_bytes, h->stream, a);
    } else {
        if (compact)
            hipLaunchKernelGGL((tab_step_kernel<false, true>), dim3(blocks), dim3(kTabBlock), 0, h->stream, a);
        else
            hipLaunchKernelGGL((tab_step_kernel<false, false>), dim3(blocks), dim3(kTabBlock), 0, h->stream, a);
    }
    TAB_HIP(h, hipGetLastError());
    h->last_kernel = MXV_TAB_KERNEL_GENERAL;
    return tab_clock_add(h, K);
}

// T = ceil(cum * 2^32 - 0.5) clamped to [0, 2^32]: cum > (w + 0.5) * 2^-32  <=>  w < T for every 32-bit word w (both sides of the
// rewrite are exact in fp64: cum * 2^32 is a scaling, the subtraction of 0.5 from a value below 2^33 loses nothing).
uint64_t word_threshold(double cum) {
    const double x = cum * 4294967296.0 - 0.5;
    if (!(x > 0.0)) return 0;
    if (x > 4294967295.0) return 1ull << 32;
    return (uint64_t)std::ceil(x);
}

// The packed table of tab_traj_kernel, or an empty vector when this MDP has to stay on the general kernel.
std::vector<uint32_t> pack_fast_table(const mxv_tab_config &cfg, const double *cum, const double *prob, const int32_t *next,
                                      const double *reward, const uint8_t *term, const double *init_cum, int single_start, bool all_one,
                                      bool compact, int *ent_off, int *init_off, int *S1_out, int *iters) {
    const int S = cfg.num_states, A = cfg.num_actions, M = cfg.max_transitions;
    std::vector<uint32_t> w;
    if (!(M == 1 || M == 3) || (M == 1 && !all_one)) return {};
    const size_t SA = (size_t)S * A;
    if (M > 1) w.assign(SA * 2, 0xFFFFFFFFu);
    while (w.size() % 4) w.push_back(0u);                               // the 16-byte entries are read with ds_read_b128
    *ent_off = (int)w.size();
    w.resize(w.size() + SA * (M == 1 ? 2 : 12), 0u);
    for (size_t sa = 0; sa < SA; ++sa) {
        int nvalid = 0;
        while (nvalid < M && cum[sa * M + nvalid] >= 0.0) nvalid += 1;
        if (nvalid == 0) return {};
        for (int i = nvalid; i < M; ++i)
            if (cum[sa * M + i] >= 0.0) return {};                       // padding must be a suffix
        double prev = 0.0;
        for (int i = 0; i < nvalid; ++i) {
            const double c = cum[sa * M + i];
            if (!(c >= prev)) return {};                                 // cumulative sums are non-decreasing (NaN fails too)
            prev = c;
            const uint64_t T = word_threshold(c);
            if (T == 0) return {};                                       // a head no word selects: not encodable as T - 1
            if (i == nvalid - 1 && T != (1ull << 32)) return {};         // "none exceeds u" (argmax of all-False = 0) must be impossible
            if (i < M - 1) w[sa * 2 + i] = (uint32_t)(T - 1);            // (M == 3 only: i in {0, 1})
        }
        for (int i = 0; i < nvalid; ++i) {
            const size_t k = sa * M + i;
            const float rf = (float)reward[k];
            if ((double)rf != reward[k]) return {};
            uint32_t rb;
            std::memcpy(&rb, &rf, 4);
            const uint32_t nt = (uint32_t)next[k] | (term[k] ? 0x80000000u : 0u);
            if (M == 1) {
                w[*ent_off + sa * 2] = nt;
                w[*ent_off + sa * 2 + 1] = rb;
            } else {
                uint32_t *en = &w[*ent_off + k * 4];
                en[0] = nt;
                en[1] = rb;
                if (compact) {
                    const float pf = (float)prob[k];                     // what the general kernel's (float)p stores
                    std::memcpy(&en[2], &pf, 4);
                } else {
                    std::memcpy(&en[2], &prob[k], 8);
                }
            }
        }
    }
    *init_off = (int)w.size();
    *S1_out = 0;
    *iters = 0;
    if (single_start < 0) {
        int S1 = -1;
        double prev = 0.0;
        for (int i = 0; i < S; ++i) {
            if (!(init_cum[i] >= prev)) return {};
            prev = init_cum[i];
            if (S1 < 0 && word_threshold(init_cum[i]) == (1ull << 32)) S1 = i;
        }
        if (S1 < 1) return {};                                           // never reaches 1 (S1 == 0 would be a point mass: single_start)
        for (int i = 0; i < S1; ++i) w.push_back((uint32_t)word_threshold(init_cum[i]));
        w.push_back(0xFFFFFFFFu);                                        // index S1: read (never used) when the search has converged there
        *S1_out = S1;
        while ((1 << *iters) <= S1) *iters += 1;                         // bit_length(S1) halvings settle an interval of S1 states
    }
    while (w.size() % 4) w.push_back(0u);
    if (w.size() * 4 > 64 * 1024) return {};
    return w;
}

template <int M_T, bool COMPACT, bool SINGLE>
void launch_traj(mxv_tab *h, const TabTrajArgs &a0) {
    TabTrajArgs a = a0;
    const unsigned full = (unsigned)(a.n / kTabBlock);
    const size_t lds = (size_t)a.table_words * 4;
    if (full) hipLaunchKernelGGL((tab_traj_kernel<M_T, COMPACT, SINGLE, true>), dim3(full), dim3(kTabBlock), lds, h->stream, a);
    if (a.n % kTabBlock) {
        a.tile_base = full;
        hipLaunchKernelGGL((tab_traj_kernel<M_T, COMPACT, SINGLE, false>), dim3(1), dim3(kTabBlock), lds, h->stream, a);
    }
}

int tab_launch_traj(mxv_tab *h, int K, void *actions_out, void *obs, void *reward, uint8_t *term, uint8_t *trunc, void *prob, bool compact) {
    if (!h->was_reset)
        return tfail(h, MXV_ERR_RESET_NEEDED, "Cannot call step before calling reset (gym.error.ResetNeeded)");
    if (int rc = tab_check_buffers(h, compact, nullptr, actions_out, nullptr, obs, reward, prob, nullptr, nullptr)) return rc;
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    TabTrajArgs a{};
    a.state = h->state; a.elapsed = h->elapsed; a.seeds = h->seeds;
    a.table = h->fast_tbl; a.table_words = h->fast_words; a.A = h->cfg.num_actions; a.ent_off = h->fast_ent_off;
    a.init_off = h->fast_init_off; a.S1 = h->fast_S1; a.init_iters = h->fast_iters; a.start = h->single_start;
    a.actions = (char *)actions_out; a.obs = (char *)obs; a.reward = (char *)reward; a.prob = (char *)prob;
    a.terminated = term; a.truncated = trunc;
    a.n = h->cfg.num_envs; a.env0 = (uint64_t)h->cfg.env_offset; a.base_seed = h->base_seed; a.action_seed = h->action_seed; a.t = h->dev_clock ? 0 : h->t;
    a.t_dev = h->dev_clock ? h->t_dev : nullptr;
    a.max_steps = h->cfg.max_episode_steps; a.K = K;
    a.ep_acc = h->ep_acc;
    a.ep_return_out = h->ep_acc ? h->ep_return_out : nullptr;
    a.ep_length_out = h->ep_acc ? h->ep_length_out : nullptr;
    const bool single = h->single_start >= 0;
    const int sel = (h->fast_M == 3 ? 4 : 0) | (compact ? 2 : 0) | (single ? 1 : 0);
    switch (sel) {
        case 0: launch_traj<1, false, false>(h, a); break;
        case 1: launch_traj<1, false, true>(h, a); break;
        case 2: launch_traj<1, true, false>(h, a); break;
        case 3: launch_traj<1, true, true>(h, a); break;
        case 4: launch_traj<3, false, false>(h, a); break;
        case 5: launch_traj<3, false, true>(h, a); break;
        case 6: launch_traj<3, true, false>(h, a); break;
        default: launch_traj<3, true, true>(h, a); break;
    }
    TAB_HIP(h, hipGetLastError());
    h->last_kernel = MXV_TAB_KERNEL_TRAJECTORY;
    return tab_clock_add(h, K);
}

int tab_do_reset(mxv_tab *h, const uint8_t *mask_dev, int64_t *obs_dev) {
    if (int rc = tab_aligned(h, obs_dev, 8, "obs")) return rc;
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    h->r += 1;
    TabResetArgs a{};
    a.state = h->state; a.elapsed = h->elapsed; a.seeds = h->seeds; a.mask = mask_dev; a.init_cum = h->init_cum;
    a.obs = obs_dev; a.S = h->cfg.num_states; a.log2S = h->log2S; a.n = h->cfg.num_envs;
    a.env0 = (uint64_t)h->cfg.env_offset; a.base_seed = h->base_seed; a.t = h->dev_clock ? 0 : h->t; a.t_dev = h->dev_clock ? h->t_dev : nullptr; a.r = h->r;
    a.ep_acc = h->ep_acc;
    const unsigned blocks = (unsigned)((h->cfg.num_envs + kTabBlock - 1) / kTabBlock);
    hipLaunchKernelGGL(tab_reset_kernel, dim3(blocks), dim3(kTabBlock), 0, h->stream, a);
    TAB_HIP(h, hipGetLastError());
    h->was_reset = true;
    return MXV_OK;
}

int tab_ensure_staging(mxv_tab *h) {
    if (h->st_obs) return MXV_OK;
    const size_t n = (size_t)h->cfg.num_envs;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b8 = up(n * 8), b1 = up(n), total = 6 * b8 + up(2 * n * 8) + 3 * b1 + 256;
    if (total <= (size_t)2 << 20) {
        TAB_HIP(h, hipHostMalloc((void **)&h->hm_block, total, hipHostMallocDefault));
        char *p = h->hm_block;
        h->st_actions = (int64_t *)p; p += b8;
        h->st_obs = (int64_t *)p; p += b8;
        h->st_final = (int64_t *)p; p += b8;
        h->st_reward = (double *)p; p += b8;
        h->st_prob = (double *)p; p += b8;
        h->st_fprob = (double *)p; p += b8;
        h->st_uniforms = (double *)p; p += up(2 * n * 8);
        h->st_term = (uint8_t *)p; p += b1;
        h->st_trunc = (uint8_t *)p; p += b1;
        h->st_mask = (uint8_t *)p; p += b1;
        h->hm_err = (int32_t *)p;
        *h->hm_err = 0;
        h->hostmap = true;
        return MXV_OK;
    }
    TAB_HIP(h, hipMalloc((void **)&h->st_actions, n * 8));
    TAB_HIP(h, hipMalloc((void **)&h->st_obs, n * 8));
    TAB_HIP(h, hipMalloc((void **)&h->st_final, n * 8));
    TAB_HIP(h, hipMalloc((void **)&h->st_reward, n * 8));
    TAB_HIP(h, hipMalloc((void **)&h->st_prob, n * 8));
    TAB_HIP(h, hipMalloc((void **)&h->st_fprob, n * 8));
    TAB_HIP(h, hipMalloc((void **)&h->st_uniforms, 2 * n * 8));
    TAB_HIP(h, hipMalloc((void **)&h->st_term, n));
    TAB_HIP(h, hipMalloc((void **)&h->st_trunc, n));
    TAB_HIP(h, hipMalloc((void **)&h->st_mask, n));
    return MXV_OK;
}

}  // namespace

extern "C" {

int mxv_tab_create(const mxv_tab_config *cfg, const double *cum_prob_host, const double *prob_host,
                   const int32_t *next_state_host, const double *reward_host, const uint8_t *terminated_host,
                   const double *initial_cum_host, mxv_tab **out) {
    if (!cfg || !out) return tfail(nullptr, MXV_ERR_INVALID_ARG, "NULL config or output pointer");
    *out = nullptr;
    if (!cum_prob_host || !prob_host || !next_state_host || !reward_host || !terminated_host || !initial_cum_host)
        return tfail(nullptr, MXV_ERR_INVALID_ARG, "NULL table pointer");
    const int S = cfg->num_states, A = cfg->num_actions, M = cfg->max_transitions;
    if (S <= 0 || A <= 0 || M <= 0 || (int64_t)S * A * M > (1 << 26))
        return tfail(nullptr, MXV_ERR_INVALID_ARG, "bad table dimensions S=%d A=%d M=%d", S, A, M);
    if (cfg->num_envs <= 0) return tfail(nullptr, MXV_ERR_INVALID_ARG, "num_envs must be positive");
    if (cfg->env_offset < 0 || cfg->env_offset % MXV_ENV_ALIGN != 0)
        return tfail(nullptr, MXV_ERR_INVALID_ARG, "env_offset must be a non-negative multiple of %d", MXV_ENV_ALIGN);
    const size_t entries = (size_t)S * A * M;
    for (size_t i = 0; i < entries; ++i)
        if (next_state_host[i] < 0 || next_state_host[i] >= S)
            return tfail(nullptr, MXV_ERR_INVALID_ARG, "next_state[%zu] = %d outside [0, %d)", i, next_state_host[i], S);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return tfail(nullptr, MXV_ERR_HIP, "no HIP device available (%s): the engine has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (cfg->device < 0 || cfg->device >= ndev) return tfail(nullptr, MXV_ERR_INVALID_ARG, "device %d out of range", cfg->device);
    mxv_tab *h = new (std::nothrow) mxv_tab();
    if (!h) return tfail(nullptr, MXV_ERR_INVALID_ARG, "out of host memory");
    h->cfg = *cfg;
    h->base_seed = cfg->seed;
    h->action_seed = cfg->action_seed;
    while ((1 << h->log2S) < S) h->log2S += 1;
    // categorical_sample over a point mass returns that state for every u in [0, 1): cumulative sums 0,...,0,1,...,1
    for (int i = 0; i < S; ++i) {
        const double prev = i ? initial_cum_host[i - 1] : 0.0;
        if (prev == 0.0 && initial_cum_host[i] == 1.0 && initial_cum_host[S - 1] == 1.0) h->single_start = i;
        if (initial_cum_host[i] != 0.0) break;
    }
    bool all_one = true;
    for (size_t i = 0; i < entries; ++i)
        if (cum_prob_host[i] >= 0.0 && prob_host[i] != 1.0) all_one = false;
    const bool need_cum = M > 1;
    std::vector<int32_t> packed(entries);
    for (size_t i = 0; i < entries; ++i) packed[i] = next_state_host[i] | (terminated_host[i] ? (int32_t)0x80000000 : 0);
    const size_t n = (size_t)cfg->num_envs;
#define TAB_CREATE_HIP(expr)                                                     \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) {                                                  \
            tfail(nullptr, MXV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
            mxv_tab_destroy(h);                                                  \
            return MXV_ERR_HIP;                                                  \
        }                                                                        \
    } while (0)
    TAB_CREATE_HIP(hipSetDevice(cfg->device));
    TAB_CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    TAB_CREATE_HIP(hipMalloc((void **)&h->state, n * sizeof(int32_t)));
    TAB_CREATE_HIP(hipMalloc((void **)&h->elapsed, n * sizeof(int32_t)));
    TAB_CREATE_HIP(hipMalloc((void **)&h->err, sizeof(int32_t)));
    TAB_CREATE_HIP(hipMalloc((void **)&h->t_dev, sizeof(uint64_t)));
    TAB_CREATE_HIP(hipMalloc((void **)&h->reward, entries * sizeof(double)));
    TAB_CREATE_HIP(hipMalloc((void **)&h->nt, entries * sizeof(int32_t)));
    TAB_CREATE_HIP(hipMalloc((void **)&h->init_cum, (size_t)S * sizeof(double)));
    if (need_cum) TAB_CREATE_HIP(hipMalloc((void **)&h->cum, entries * sizeof(double)));
    if (!all_one) TAB_CREATE_HIP(hipMalloc((void **)&h->prob, entries * sizeof(double)));
    TAB_CREATE_HIP(hipMemsetAsync(h->state, 0, n * sizeof(int32_t), h->stream));
    TAB_CREATE_HIP(hipMemsetAsync(h->elapsed, 0, n * sizeof(int32_t), h->stream));
    TAB_CREATE_HIP(hipMemsetAsync(h->err, 0, sizeof(int32_t), h->stream));
    TAB_CREATE_HIP(hipMemsetAsync(h->t_dev, 0, sizeof(uint64_t), h->stream));
    TAB_CREATE_HIP(hipMemcpyAsync(h->reward, reward_host, entries * sizeof(double), hipMemcpyHostToDevice, h->stream));
    TAB_CREATE_HIP(hipMemcpyAsync(h->nt, packed.data(), entries * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    TAB_CREATE_HIP(hipMemcpyAsync(h->init_cum, initial_cum_host, (size_t)S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (need_cum) TAB_CREATE_HIP(hipMemcpyAsync(h->cum, cum_prob_host, entries * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (!all_one) TAB_CREATE_HIP(hipMemcpyAsync(h->prob, prob_host, entries * sizeof(double), hipMemcpyHostToDevice, h->stream));
    TAB_CREATE_HIP(hipStreamSynchronize(h->stream));
#undef TAB_CREATE_HIP
    h->lds_bytes = entries * (sizeof(double) * (1 + (need_cum ? 1 : 0) + (all_one ? 0 : 1)) + sizeof(int32_t)) + (size_t)S * sizeof(double);
    h->lds_table = h->lds_bytes <= 64 * 1024;  // larger MDPs (custom maps) read the table through L2 instead
    if (!(cfg->flags & MXV_TAB_FLAG_GENERAL_KERNEL)) {
        const std::vector<uint32_t> w = pack_fast_table(*cfg, cum_prob_host, prob_host, next_state_host, reward_host, terminated_host,
                                                        initial_cum_host, h->single_start, all_one, (cfg->flags & MXV_TAB_FLAG_COMPACT) != 0,
                                                        &h->fast_ent_off, &h->fast_init_off, &h->fast_S1, &h->fast_iters);
        if (!w.empty()) {
            hipError_t e1 = hipMalloc((void **)&h->fast_tbl, w.size() * 4);
            if (e1 == hipSuccess) e1 = hipMemcpy(h->fast_tbl, w.data(), w.size() * 4, hipMemcpyHostToDevice);
            if (e1 != hipSuccess) {
                tfail(nullptr, MXV_ERR_HIP, "packed transition table: %s", hipGetErrorString(e1));
                mxv_tab_destroy(h);
                return MXV_ERR_HIP;
            }
            h->fast_M = M;
            h->fast_words = (int)w.size();
        }
    }
    *out = h;
    return MXV_OK;
}

int mxv_tab_destroy(mxv_tab *h) {
    if (!h) return MXV_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->state, h->elapsed, h->err, h->nt, h->seeds, h->cum, h->prob, h->reward, h->init_cum, h->fast_tbl, h->t_dev,
                    h->ep_acc, h->st_ep_r, h->st_ep_l};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->hostmap) {
        (void)hipHostFree(h->hm_block);
    } else {
        void *stage[] = {h->st_actions, h->st_obs, h->st_final, h->st_reward, h->st_prob, h->st_fprob, h->st_uniforms, h->st_term,
                         h->st_trunc, h->st_mask};
        for (void *p : stage)
            if (p) (void)hipFree(p);
    }
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MXV_OK;
}

const char *mxv_tab_last_error(const mxv_tab *h) { return h ? h->error.c_str() : g_tab_create_error.c_str(); }

int mxv_tab_seed(mxv_tab *h, uint64_t base_seed, const uint64_t *per_env_seeds_host) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    h->base_seed = base_seed;
    h->t = 0;
    h->r = 0;
    if (int rc = tab_clock_set(h)) return rc;
    if (per_env_seeds_host) {
        const size_t bytes = (size_t)h->cfg.num_envs * sizeof(uint64_t);
        if (!h->seeds) TAB_HIP(h, hipMalloc((void **)&h->seeds, bytes));
        TAB_HIP(h, hipMemcpyAsync(h->seeds, per_env_seeds_host, bytes, hipMemcpyHostToDevice, h->stream));
        TAB_HIP(h, hipStreamSynchronize(h->stream));
    } else if (h->seeds) {
        TAB_HIP(h, hipFree(h->seeds));
        h->seeds = nullptr;
    }
    return MXV_OK;
}

int mxv_tab_seed_actions(mxv_tab *h, uint64_t action_seed) {
    TAB_CHECK(h);
    h->action_seed = action_seed;
    return MXV_OK;
}

int mxv_tab_reset(mxv_tab *h, const uint8_t *mask_dev, int64_t *obs_dev) {
    TAB_CHECK(h);
    return tab_do_reset(h, mask_dev, obs_dev);
}

int mxv_tab_step(mxv_tab *h, const int64_t *actions_dev, const double *uniforms_dev, int64_t *obs_dev, double *reward_dev,
                 uint8_t *terminated_dev, uint8_t *truncated_dev, double *prob_dev, int64_t *final_obs_dev,
                 double *final_prob_dev) {
    TAB_CHECK(h);
    if (!actions_dev) return tfail(h, MXV_ERR_INVALID_ARG, "actions pointer is NULL (use mxv_tab_rollout for sampled actions)");
    return tab_launch(h, 1, 0, actions_dev, 0, nullptr, uniforms_dev, obs_dev, reward_dev, terminated_dev, truncated_dev,
                      prob_dev, final_obs_dev, final_prob_dev);
}

int mxv_tab_rollout(mxv_tab *h, int32_t K, int32_t per_step, void *actions_out_dev, void *obs_dev, void *reward_dev,
                    uint8_t *terminated_dev, uint8_t *truncated_dev, void *prob_dev, void *final_obs_dev,
                    void *final_prob_dev) {
    TAB_CHECK(h);
    const bool compact = (h->cfg.flags & MXV_TAB_FLAG_COMPACT) != 0;
    if (h->fast_M != 0 && per_step && K > 0 && actions_out_dev && obs_dev && reward_dev && terminated_dev && truncated_dev && prob_dev &&
        !final_obs_dev && !final_prob_dev)
        return tab_launch_traj(h, K, actions_out_dev, obs_dev, reward_dev, terminated_dev, truncated_dev, prob_dev, compact);
    return tab_launch(h, K, per_step ? h->cfg.num_envs : 0, nullptr, 0, actions_out_dev, nullptr, obs_dev, reward_dev,
                      terminated_dev, truncated_dev, prob_dev, final_obs_dev, final_prob_dev, compact);
}

int mxv_tab_rollout_tape(mxv_tab *h, int32_t K, int32_t per_step, const void *actions_tape_dev, void *obs_dev,
                         void *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, void *prob_dev,
                         void *final_obs_dev, void *final_prob_dev) {
    TAB_CHECK(h);
    if (!actions_tape_dev) return tfail(h, MXV_ERR_INVALID_ARG, "actions tape pointer is NULL");
    return tab_launch(h, K, per_step ? h->cfg.num_envs : 0, actions_tape_dev, h->cfg.num_envs, nullptr, nullptr, obs_dev,
                      reward_dev, terminated_dev, truncated_dev, prob_dev, final_obs_dev, final_prob_dev,
                      (h->cfg.flags & MXV_TAB_FLAG_COMPACT) != 0);
}

int mxv_tab_reset_host(mxv_tab *h, const uint8_t *mask_host, int64_t *obs_host) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = tab_ensure_staging(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (h->hostmap) {
        if (mask_host) std::memcpy(h->st_mask, mask_host, n);
        if (int rc = tab_do_reset(h, mask_host ? h->st_mask : nullptr, obs_host ? h->st_obs : nullptr)) return rc;
        TAB_HIP(h, hipStreamSynchronize(h->stream));
        if (obs_host) std::memcpy(obs_host, h->st_obs, n * 8);
        return MXV_OK;
    }
    if (mask_host) TAB_HIP(h, hipMemcpyAsync(h->st_mask, mask_host, n, hipMemcpyHostToDevice, h->stream));
    if (int rc = tab_do_reset(h, mask_host ? h->st_mask : nullptr, obs_host ? h->st_obs : nullptr)) return rc;
    if (obs_host) TAB_HIP(h, hipMemcpyAsync(obs_host, h->st_obs, n * 8, hipMemcpyDeviceToHost, h->stream));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_tab_step_host(mxv_tab *h, const int64_t *actions_host, const double *uniforms_host, int64_t *obs_host,
                      double *reward_host, uint8_t *terminated_host, uint8_t *truncated_host, double *prob_host,
                      int64_t *final_obs_host, double *final_prob_host) {
    TAB_CHECK(h);
    if (!actions_host || !obs_host) return tfail(h, MXV_ERR_INVALID_ARG, "actions/obs pointer is NULL");
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    if (int rc = tab_ensure_staging(h)) return rc;
    const size_t n = (size_t)h->cfg.num_envs;
    if (h->hostmap) {
        std::memcpy(h->st_actions, actions_host, n * 8);
        if (uniforms_host) std::memcpy(h->st_uniforms, uniforms_host, 2 * n * 8);
        h->err_in_block = true;
        h->ep_host_step = true;
        const int lrc = tab_launch(h, 1, 0, h->st_actions, 0, nullptr, uniforms_host ? h->st_uniforms : nullptr, h->st_obs,
                                   reward_host ? h->st_reward : nullptr, terminated_host ? h->st_term : nullptr,
                                   truncated_host ? h->st_trunc : nullptr, prob_host ? h->st_prob : nullptr,
                                   final_obs_host ? h->st_final : nullptr, final_prob_host ? h->st_fprob : nullptr);
        h->err_in_block = false;
        h->ep_host_step = false;
        if (lrc) return lrc;
        TAB_HIP(h, hipStreamSynchronize(h->stream));
        std::memcpy(obs_host, h->st_obs, n * 8);
        if (reward_host) std::memcpy(reward_host, h->st_reward, n * 8);
        if (terminated_host) std::memcpy(terminated_host, h->st_term, n);
        if (truncated_host) std::memcpy(truncated_host, h->st_trunc, n);
        if (prob_host) std::memcpy(prob_host, h->st_prob, n * 8);
        if (final_obs_host) std::memcpy(final_obs_host, h->st_final, n * 8);
        if (final_prob_host) std::memcpy(final_prob_host, h->st_fprob, n * 8);
        if (*h->hm_err != 0) {
            *h->hm_err = 0;
            (void)tab_clock_add(h, -1);
            return tfail(h, MXV_ERR_INVALID_ACTION, "discrete action outside [0, %d) (Discrete.contains)", h->cfg.num_actions);
        }
        return MXV_OK;
    }
    TAB_HIP(h, hipMemcpyAsync(h->st_actions, actions_host, n * 8, hipMemcpyHostToDevice, h->stream));
    if (uniforms_host) TAB_HIP(h, hipMemcpyAsync(h->st_uniforms, uniforms_host, 2 * n * 8, hipMemcpyHostToDevice, h->stream));
    h->ep_host_step = true;
    const int lrc = tab_launch(h, 1, 0, h->st_actions, 0, nullptr, uniforms_host ? h->st_uniforms : nullptr, h->st_obs,
                               reward_host ? h->st_reward : nullptr, terminated_host ? h->st_term : nullptr,
                               truncated_host ? h->st_trunc : nullptr, prob_host ? h->st_prob : nullptr,
                               final_obs_host ? h->st_final : nullptr, final_prob_host ? h->st_fprob : nullptr);
    h->ep_host_step = false;
    if (lrc) return lrc;
    TAB_HIP(h, hipMemcpyAsync(obs_host, h->st_obs, n * 8, hipMemcpyDeviceToHost, h->stream));
    if (reward_host) TAB_HIP(h, hipMemcpyAsync(reward_host, h->st_reward, n * 8, hipMemcpyDeviceToHost, h->stream));
    if (terminated_host) TAB_HIP(h, hipMemcpyAsync(terminated_host, h->st_term, n, hipMemcpyDeviceToHost, h->stream));
    if (truncated_host) TAB_HIP(h, hipMemcpyAsync(truncated_host, h->st_trunc, n, hipMemcpyDeviceToHost, h->stream));
    if (prob_host) TAB_HIP(h, hipMemcpyAsync(prob_host, h->st_prob, n * 8, hipMemcpyDeviceToHost, h->stream));
    if (final_obs_host) TAB_HIP(h, hipMemcpyAsync(final_obs_host, h->st_final, n * 8, hipMemcpyDeviceToHost, h->stream));
    if (final_prob_host) TAB_HIP(h, hipMemcpyAsync(final_prob_host, h->st_fprob, n * 8, hipMemcpyDeviceToHost, h->stream));
    int rc = tab_check_latched(h);
    if (rc == MXV_ERR_INVALID_ACTION) (void)tab_clock_add(h, -1);
    return rc;
}

/* gym.wrappers.RecordEpisodeStatistics fused into the step (record_episode_statistics.py:96-151): see mxv_toytext.h */
int mxv_tab_episode_stats(mxv_tab *h, int32_t enable) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->cfg.num_envs;
    if (enable && !h->ep_acc) {
        TAB_HIP(h, hipMalloc((void **)&h->ep_acc, n * sizeof(float)));
        TAB_HIP(h, hipMalloc((void **)&h->st_ep_r, n * sizeof(float)));
        TAB_HIP(h, hipMalloc((void **)&h->st_ep_l, n * sizeof(int32_t)));
        TAB_HIP(h, hipMemsetAsync(h->ep_acc, 0, n * sizeof(float), h->stream));
        TAB_HIP(h, hipMemsetAsync(h->st_ep_r, 0, n * sizeof(float), h->stream));
        TAB_HIP(h, hipMemsetAsync(h->st_ep_l, 0, n * sizeof(int32_t), h->stream));
        TAB_HIP(h, hipStreamSynchronize(h->stream));
    } else if (!enable && h->ep_acc) {
        TAB_HIP(h, hipFree(h->ep_acc));
        TAB_HIP(h, hipFree(h->st_ep_r));
        TAB_HIP(h, hipFree(h->st_ep_l));
        h->ep_acc = h->st_ep_r = nullptr;
        h->st_ep_l = nullptr;
    }
    return MXV_OK;
}

int mxv_tab_set_episode_outputs(mxv_tab *h, float *ep_return_dev, int32_t *ep_length_dev) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    h->ep_return_out = ep_return_dev;
    h->ep_length_out = ep_length_dev;
    return MXV_OK;
}

int mxv_tab_episode_stats_host(mxv_tab *h, float *ep_return_host, int32_t *ep_length_host, float *running_return_host) {
    TAB_CHECK(h);
    if (!h->ep_acc) return tfail(h, MXV_ERR_INVALID_ARG, "episode statistics are not enabled (mxv_tab_episode_stats)");
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->cfg.num_envs;
    if (ep_return_host) TAB_HIP(h, hipMemcpyAsync(ep_return_host, h->st_ep_r, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (ep_length_host) TAB_HIP(h, hipMemcpyAsync(ep_length_host, h->st_ep_l, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (running_return_host) TAB_HIP(h, hipMemcpyAsync(running_return_host, h->ep_acc, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_tab_set_running_returns(mxv_tab *h, const float *running_return_host) {
    TAB_CHECK(h);
    if (!h->ep_acc) return tfail(h, MXV_ERR_INVALID_ARG, "episode statistics are not enabled (mxv_tab_episode_stats)");
    if (!running_return_host) return tfail(h, MXV_ERR_INVALID_ARG, "NULL pointer");
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    TAB_HIP(h, hipMemcpyAsync(h->ep_acc, running_return_host, (size_t)h->cfg.num_envs * sizeof(float), hipMemcpyHostToDevice, h->stream));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_tab_get_state(mxv_tab *h, int32_t *state_host, int32_t *elapsed_host) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->cfg.num_envs;
    if (state_host) TAB_HIP(h, hipMemcpyAsync(state_host, h->state, n * 4, hipMemcpyDeviceToHost, h->stream));
    if (elapsed_host) TAB_HIP(h, hipMemcpyAsync(elapsed_host, h->elapsed, n * 4, hipMemcpyDeviceToHost, h->stream));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    return MXV_OK;
}

int mxv_tab_set_state(mxv_tab *h, const int32_t *state_host, const int32_t *elapsed_host) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->cfg.num_envs;
    if (state_host) {
        for (size_t i = 0; i < n; ++i)
            if (state_host[i] < 0 || state_host[i] >= h->cfg.num_states)
                return tfail(h, MXV_ERR_INVALID_ARG, "state[%zu] = %d outside [0, %d)", i, state_host[i], h->cfg.num_states);
        TAB_HIP(h, hipMemcpyAsync(h->state, state_host, n * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (elapsed_host) TAB_HIP(h, hipMemcpyAsync(h->elapsed, elapsed_host, n * 4, hipMemcpyHostToDevice, h->stream));
    TAB_HIP(h, hipStreamSynchronize(h->stream));
    h->was_reset = true;
    return MXV_OK;
}

int mxv_tab_get_counters(mxv_tab *h, uint64_t *t, uint32_t *r) {
    TAB_CHECK(h);
    if (h->dev_clock) {  // a caller's graph replays advance the device word only
        TAB_HIP(h, hipSetDevice(h->cfg.device));
        TAB_HIP(h, hipMemcpyAsync(&h->t, h->t_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        TAB_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (t) *t = h->t;
    if (r) *r = h->r;
    return MXV_OK;
}

int mxv_tab_set_counters(mxv_tab *h, uint64_t t, uint32_t r) {
    TAB_CHECK(h);
    h->t = t;
    h->r = r;
    return tab_clock_set(h);
}

int mxv_tab_set_device_clock(mxv_tab *h, int32_t on) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    if (on && !h->dev_clock) {
        h->dev_clock = true;
        return tab_clock_set(h);
    }
    if (!on && h->dev_clock) {
        TAB_HIP(h, hipMemcpyAsync(&h->t, h->t_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        TAB_HIP(h, hipStreamSynchronize(h->stream));
        h->dev_clock = false;
    }
    return MXV_OK;
}

uint64_t mxv_tab_word_threshold(double cum_prob) { return word_threshold(cum_prob); }

int mxv_tab_last_kernel(const mxv_tab *h) { return h ? h->last_kernel : MXV_TAB_KERNEL_NONE; }

int mxv_tab_sync(mxv_tab *h) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    return tab_check_latched(h);
}

int mxv_tab_set_stream(mxv_tab *h, void *stream) {
    TAB_CHECK(h);
    TAB_HIP(h, hipSetDevice(h->cfg.device));
    if (h->stream) TAB_HIP(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) TAB_HIP(h, hipStreamDestroy(h->stream));
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return MXV_OK;
}

}  // extern "C"
