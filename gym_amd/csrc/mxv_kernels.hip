// mxv_kernels.hip — hand-written gfx950 kernels of the vectorised classic-control engine.
//
// step_kernel: ONE launch = K x SyncVectorEnv.step_wait (gym/vector/sync_vector_env.py:135-169) for
// every env of the shard: dynamics + TimeLimit (gym/wrappers/time_limit.py:50-54) + termination
// + autoreset, optionally with the actions drawn on device (Philox action stream); K > 1 fuses a
// rollout chunk into one launch with the env state resident in registers between steps.
//
// Mapping.  A workgroup of 256 lanes (4 wave64) owns a tile of E*256 consecutive envs; lane `tid`
// owns envs tile0 + j*256 + tid, j < E.  For every j a wave touches 64 consecutive elements of
// each struct-of-arrays state component (512 B of fp64), of elapsed[], reward[], the flag arrays
// and 64 consecutive observation rows — every global access is a fully coalesced, line-aligned
// burst, and the E independent env chains per lane give the VALU instruction-level parallelism
// while E*(S+1) loads per lane are in flight.  No MFMA: this is element-wise fp64 physics.
// LDS is used only to transpose the Philox action words: one Philox call yields the words of 4
// consecutive envs (group g = env>>2), which belong to 4 different lanes under the mapping above.
#include "mxv_kernels.hpp"

namespace mxv {

namespace {

template <int O>
__device__ __forceinline__ void store_obs(float *base, int64_t e, const float *o) {
    if constexpr (O == 4) {
        reinterpret_cast<float4 *>(base)[e] = make_float4(o[0], o[1], o[2], o[3]);
    } else if constexpr (O == 2) {
        reinterpret_cast<float2 *>(base)[e] = make_float2(o[0], o[1]);
    } else if constexpr (O == 6) {
        float2 *p = reinterpret_cast<float2 *>(base) + e * 3;
        p[0] = make_float2(o[0], o[1]);
        p[1] = make_float2(o[2], o[3]);
        p[2] = make_float2(o[4], o[5]);
    } else {
        float *p = base + e * O;
#pragma unroll
        for (int k = 0; k < O; ++k) p[k] = o[k];
    }
}

// XCD-aware workgroup -> tile map.  The hardware deals workgroup ids round-robin over the 8 XCDs (id % 8), each with
// its own L2.  Handing out tiles in id order makes every XCD write 4-8 KiB crumbs interleaved with the other seven
// all over each output row; giving XCD x the x-th contiguous eighth of the tiles instead lets each L2 stream long
// contiguous runs to its memory channels.  Measured on the rollout's store pattern with the physics removed
// (tools/wbench, profiles/r01_wbench.txt): 4.7 -> 5.7 TB/s.  Works for any tile count (remainder tiles go to the
// low XCDs, matching how many ids of each residue exist).
constexpr unsigned kXcds = 8;
__device__ __forceinline__ unsigned xcd_contiguous_tile(unsigned bid, unsigned ntiles) {
#if MXV_XCD_MAP
    const unsigned x = bid % kXcds, idx = bid / kXcds;
#if MXV_XCD_BLOCK > 0
    // XCDs take turns in blocks of MXV_XCD_BLOCK tiles (tuning variant; tiles past the end are skipped by the callers' bounds)
    if (ntiles % (kXcds * MXV_XCD_BLOCK) == 0) return ((idx / MXV_XCD_BLOCK) * kXcds + x) * MXV_XCD_BLOCK + idx % MXV_XCD_BLOCK;
#endif
    const unsigned base = ntiles / kXcds, rem = ntiles % kXcds;
    return x * base + (x < rem ? x : rem) + idx;
#else
    return bid;
#endif
}

// Word `idx` (0..3, runtime) of a Philox result.
[[maybe_unused]] __device__ __forceinline__ uint32_t pick_word(const U4 &w, uint32_t idx) {
    return idx == 0 ? w.x : (idx == 1 ? w.y : (idx == 2 ? w.z : w.w));
}

// step_kernel<ENV, DEF, E, CONSEC>: a.K vector steps in ONE launch, env state held in registers between
// steps (K = 1 is the plain step() call).  Per step the only HBM traffic is the step's outputs; state
// and elapsed[] are read once at entry and written once at exit, i.e. 16*S/K + 8/K bytes per env-step.
//   CONSEC = false: lane owns envs tile0 + j*256 + tid (every access of a wave is a dense burst);
//                   Philox action words are transposed through LDS (one call = 4 consecutive envs).
//   CONSEC = true : lane owns the E consecutive envs tile0 + tid*E + j: a Philox action group is
//                   lane-private (no LDS, no barrier) and the flag bytes of a lane are contiguous.
template <int ENV, int DEF, int E, bool CONSEC, bool MULTI>
__global__ void __launch_bounds__(kBlock, MXV_MIN_WAVES) step_kernel(const StepArgs a) {
    using EV = Env<ENV>;
    constexpr int S = EV::S, O = EV::O, NA = EV::NA;
    constexpr int TILE = E * kBlock;
    static_assert(!CONSEC || E == 1 || E == 2 || E % 4 == 0, "CONSEC needs E in {1, 2, 4k}");
    const int tid = threadIdx.x;
    const int64_t tile0 = (int64_t)xcd_contiguous_tile(blockIdx.x, gridDim.x) * TILE;
    const int64_t n = a.n;
    const Par<DEF> P(a.P, a.params_pe, a.n);
    const uint64_t t0 = a.t + (a.t_dev ? *a.t_dev : 0);
    const bool autoreset = !(a.flags & MXV_FLAG_NO_AUTORESET);
    const bool sampled = a.actions == nullptr;
    auto env_of = [&](int j) -> int64_t { return CONSEC ? tile0 + (int64_t)tid * E + j : tile0 + (int64_t)j * kBlock + tid; };

    // ---- entry: state + elapsed of the lane's E envs ----
    constexpr int AUXN = EV::AUX > 0 ? EV::AUX : 1;
    double s[E][S], aux[E][AUXN];
    int32_t el[E];
    bool valid[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t e = env_of(j);
        valid[j] = e < n;
        const int64_t ec = valid[j] ? e : 0;
#pragma unroll
        for (int k = 0; k < S; ++k) s[j][k] = a.state[(int64_t)k * n + ec];
        el[j] = a.elapsed[ec];
        EV::prime(s[j], aux[j]);
    }
    float er[E];  // running episode return (RecordEpisodeStatistics.episode_returns)
#pragma unroll
    for (int j = 0; j < E; ++j) er[j] = a.ep_acc ? a.ep_acc[valid[j] ? env_of(j) : 0] : 0.0f;
    __shared__ uint32_t sw[CONSEC ? 4 : TILE];

    const int nsteps = MULTI ? a.K : 1;  // MULTI = false: the plain step() launch, no loop-carried bookkeeping
    if (MULTI) settle_entry_loads();
    for (int step = 0; step < nsteps; ++step) {
        const uint64_t t = t0 + (uint64_t)step;
        const int64_t so = MULTI ? (int64_t)step * a.slice : 0;      // output slice offset (envs): [K][N] trajectories or 0
        const int64_t sa = MULTI ? (int64_t)step * a.act_slice : 0;  // action tape offset

        // ---- actions ----
        int ai[E];
        float af[E];
        if (sampled) {
            if constexpr (CONSEC) {
#pragma unroll
                for (int j0 = 0; j0 < E; j0 += 4) {
                    const uint64_t ge = a.env0 + (uint64_t)env_of(j0);
                    const U4 w = action_words(a.action_seed, t, ge >> 2);
#pragma unroll
                    for (int q = 0; q < 4 && j0 + q < E; ++q)
                        action_from_word<ENV, DEF>(P.at(valid[j0 + q] ? env_of(j0 + q) : 0), pick_word(w, (uint32_t)((ge + q) & 3)), ai[j0 + q], af[j0 + q]);
                }
            } else {
                // thread c computes the 4 words of group (env0 + tile0)/4 + c; LDS hands them to the owning lanes
                if (MULTI && step > 0) __syncthreads();
                for (int c = tid; c < TILE / 4; c += kBlock) {
                    const uint64_t g = ((a.env0 + (uint64_t)tile0) >> 2) + (uint64_t)c;
                    const U4 w = action_words(a.action_seed, t, g);
                    reinterpret_cast<uint4 *>(sw)[c] = make_uint4(w.x, w.y, w.z, w.w);
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < E; ++j) action_from_word<ENV, DEF>(P.at(valid[j] ? env_of(j) : 0), sw[j * kBlock + tid], ai[j], af[j]);
            }
            if (a.actions_out != nullptr) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    if (!valid[j]) continue;
                    const int64_t e = so + env_of(j);
                    if constexpr (NA > 0) {
                        if (a.flags & MXV_FLAG_ACTION_I32)
                            static_cast<int32_t *>(a.actions_out)[e] = ai[j];
                        else
                            static_cast<int64_t *>(a.actions_out)[e] = ai[j];
                    } else {
                        static_cast<float *>(a.actions_out)[e] = af[j];
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int64_t ec = sa + (valid[j] ? env_of(j) : 0);
                if constexpr (NA > 0) {
                    const int64_t v = (a.flags & MXV_FLAG_ACTION_I32)
                                          ? (int64_t) static_cast<const int32_t *>(a.actions)[ec]
                                          : static_cast<const int64_t *>(a.actions)[ec];
                    // Discrete.contains (cartpole.py:131-132): out of range -> latch, leave the env untouched
                    if (valid[j] && (v < 0 || v >= NA)) {
                        atomicOr(a.err, 1);
                        valid[j] = false;
                    }
                    ai[j] = (int)v;
                    af[j] = 0.0f;
                } else {
                    ai[j] = 0;
                    af[j] = static_cast<const float *>(a.actions)[ec];
                }
            }
        }

        if constexpr (ENV == MXV_ACROBOT && DEF != PM_DEFAULT) {  // torque noise: one draw per env-step when the attribute is set
            if (a.step_noise) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const int64_t e = valid[j] ? env_of(j) : 0;
                    const uint64_t seed = a.seeds ? landed(a.seeds[e]) : a.base_seed + a.env0 + (uint64_t)e;
                    af[j] = __uint_as_float(step_noise_word(seed, t));
                }
            }
        }

        // ---- dynamics + TimeLimit, E independent chains ----
        float obs[E][O];
        double rew[E];
        bool term[E], trunc[E], pend[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            term[j] = EV::template step<DEF>(P.at(valid[j] ? env_of(j) : 0), s[j], aux[j], el[j] == 0, ai[j], af[j], rew[j], obs[j]);
            el[j] += 1;                                              // time_limit.py:51
            trunc[j] = (a.max_steps > 0) && (el[j] >= a.max_steps);  // time_limit.py:53-54
            pend[j] = autoreset && (term[j] || trunc[j]);
        }
        if (a.ep_acc != nullptr) {  // record_episode_statistics.py:119-143
#pragma unroll
            for (int j = 0; j < E; ++j) {
                er[j] = (float)((double)er[j] + rew[j]);  // float32 array += float64 rewards
                if (valid[j] && (term[j] || trunc[j])) {
                    if (a.ep_return_out) a.ep_return_out[so + env_of(j)] = er[j];
                    if (a.ep_length_out) a.ep_length_out[so + env_of(j)] = el[j];
                    er[j] = 0.0f;
                }
            }
        }

        // ---- autoreset (sync_vector_env.py:152-156), compacted: every pass each lane resets its first
        // pending env, so a wave spends max-over-lanes(#finished) Philox calls, not E ----
        while (true) {
            int jsel = -1;
#pragma unroll
            for (int j = E - 1; j >= 0; --j)
                if (pend[j]) jsel = j;
            if (!__any(jsel >= 0)) break;
            if (jsel >= 0) {
                const int64_t e = CONSEC ? tile0 + (int64_t)tid * E + jsel : tile0 + (int64_t)jsel * kBlock + tid;
                bool v = false;
                float cur[O];
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (j == jsel) {
                        v = valid[j];
#pragma unroll
                        for (int k = 0; k < O; ++k) cur[k] = obs[j][k];
                    }
                if (a.final_obs != nullptr && v) store_obs<O>(a.final_obs, so + e, cur);  // info["final_observation"]
                const uint64_t seed = a.seeds ? landed(a.seeds[v ? e : 0]) : a.base_seed + a.env0 + (uint64_t)e;
                const U4 w = reset_words(seed, t, 0u);
                double ns[S], naux[AUXN];
                float nobs[O];
                EV::reset(w, a.b0, a.b1, ns);
                EV::observe(ns, nobs, naux);
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (j == jsel) {
#pragma unroll
                        for (int k = 0; k < EV::AUX; ++k) aux[j][k] = naux[k];
#pragma unroll
                        for (int k = 0; k < S; ++k) s[j][k] = ns[k];
#pragma unroll
                        for (int k = 0; k < O; ++k) obs[j][k] = nobs[k];
                        el[j] = 0;  // time_limit.py:67
                        pend[j] = false;
                    }
            }
        }

        // ---- this step's outputs ----
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (!valid[j]) continue;
            const int64_t e = so + env_of(j);
            store_obs<O>(a.obs, e, obs[j]);
            if (a.reward != nullptr) {
                if (a.flags & MXV_FLAG_REWARD_F32)
                    static_cast<float *>(a.reward)[e] = (float)rew[j];
                else
                    static_cast<double *>(a.reward)[e] = rew[j];
            }
            if (a.terminated != nullptr) a.terminated[e] = term[j] ? 1 : 0;
            if (a.truncated != nullptr) a.truncated[e] = trunc[j] ? 1 : 0;
        }
    }

    // ---- exit: state + elapsed back to HBM ----
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (!valid[j]) continue;
        const int64_t e = env_of(j);
#pragma unroll
        for (int k = 0; k < S; ++k) a.state[(int64_t)k * n + e] = s[j][k];
        a.elapsed[e] = el[j];
        if (a.ep_acc) a.ep_acc[e] = er[j];
    }
}

// ------------------------------------------------------------------------------------------------------------
// rollout_kernel<ENV, DEF, E>: the sampled-action + autoreset fast path (mxv_step_sampled, mxv_rollout FUSED/GRAPH/
// EAGER), a.K vector steps per launch with the env state in registers.
//
// Workgroup = ONE wave64 owning a tile of E*64 consecutive envs (lane L owns envs tile0 + j*64 + L): no s_barrier
// anywhere, cross-lane traffic goes through a few KiB of LDS that only this wave touches (LDS operations of one
// wave execute in order).  All Philox work is scheduled onto the 64 lanes of AT MOST ONE masked call per wave-step:
//   * the envs that finished step t are compacted with ballot/mbcnt (about 6 of 128 CartPole envs per step) and their
//     reset draws take the top lanes; the lane that draws also builds the new fp64 state AND its float32 observation,
//     so the owner only copies them back: the reset arithmetic (Pendulum: a full-range sincos) is paid once per
//     wave-step, not once per env chain;
//   * every remaining block of 16E lanes draws the action words of one FUTURE step (one call = 4 consecutive envs)
//     into a ring of 64/(16E) steps.  With nothing to reset (Pendulum, Acrobot, MountainCar between truncations) a
//     wave therefore runs Philox once every 64/(16E) steps with all 64 lanes busy, and skips the call otherwise.
// More finished envs than free lanes (e.g. every Pendulum env truncating at step 200) are handled by extra passes
// of all 64 lanes.  Output addressing: the per-step base of every array is a scalar; lanes add a 32-bit byte offset
// fixed for the whole launch, so the loop holds no 64-bit vector address arithmetic.
// ------------------------------------------------------------------------------------------------------------
constexpr int kWave = 64;

template <int S, int O, int AUX>
struct alignas(16) ResetEntry {
    double s[S];
    double x[AUX];  // Env::AUX values carried across steps
    float o[O];
};
template <int S, int O>
struct alignas(16) ResetEntry<S, O, 0> {  // no carried values: CartPole's entry is 48 B, 7.5 KiB of LDS per workgroup
    double s[S];
    float o[O];
};

template <int ENV, bool DEF, int E, bool SAFE, int OUT = 0>
__global__ void __launch_bounds__(kWave, MXV_ROLLOUT_MIN_WAVES) rollout_kernel(const StepArgs a) {
    using EV = Env<ENV>;
    constexpr int S = EV::S, O = EV::O, NA = EV::NA;
    constexpr int TILE = E * kWave;
    constexpr int NACT = TILE / 4;        // lanes that draw the action words of ONE step
    constexpr int H = kWave / NACT;       // steps of action words one full call produces = depth of the ring
    static_assert(NACT < kWave, "E must be < 4: the merged call needs free lanes");
    constexpr int AUXN = EV::AUX > 0 ? EV::AUX : 1;
    using Entry = ResetEntry<S, O, EV::AUX>;
    __shared__ uint32_t lds_act[H * TILE];  // ring of action words: slot (q % H) holds step q of this launch
    __shared__ uint32_t lds_q[TILE];      // compacted list of finished envs (tile-local index)
    __shared__ Entry lds_res[TILE];       // their new state + observation

    const int lane = threadIdx.x;
    const uint32_t tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int64_t tile0 = (int64_t)tile * TILE;
    const int64_t n = a.n;
    const Par<DEF> P(a.P);
    const uint64_t t0 = a.t + (a.t_dev ? *a.t_dev : 0);
    // OUT != 0: every per-step output array is present, no final_obs, no episode statistics, dtypes fixed (1: float64 rewards
    // + int64 actions, 2: float32 + int32) — the trajectory-recording launch.  The ~25 wave-uniform branches and ~60 scalar
    // instructions per step that the optional outputs cost fold away at compile time.
    constexpr bool FULL = OUT != 0;
    const bool act_i32 = FULL ? (OUT == 2) : ((a.flags & MXV_FLAG_ACTION_I32) != 0);
    const bool rew_f32 = FULL ? (OUT == 2) : ((a.flags & MXV_FLAG_REWARD_F32) != 0);
    const bool ep_on = !FULL && a.ep_acc != nullptr;
    const uint64_t group0 = (a.env0 + (uint64_t)tile0) >> 2;

    double s[E][S], aux[E][AUXN];
    int32_t el[E];
    bool valid[E];
    uint32_t le[E];  // env index inside the shard (fits 32 bits: mxv_create caps num_envs)
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t e = tile0 + j * kWave + lane;
        valid[j] = e < n;
        le[j] = (uint32_t)(valid[j] ? e : 0);
#pragma unroll
        for (int k = 0; k < S; ++k) s[j][k] = a.state[(int64_t)k * n + le[j]];
        el[j] = a.elapsed[le[j]];
        EV::prime(s[j], aux[j]);
    }
    float er[E];  // running episode return (RecordEpisodeStatistics.episode_returns)
#pragma unroll
    for (int j = 0; j < E; ++j) er[j] = ep_on ? a.ep_acc[le[j]] : 0.0f;
    float *p_epr = FULL ? nullptr : a.ep_return_out;
    int32_t *p_epl = FULL ? nullptr : a.ep_length_out;

    // action words of the first min(H, K) steps: lane L draws group L % NACT of step L / NACT
    int filled = a.K < H ? a.K : H;  // steps [0, filled) of this launch have their action words in the ring
    if (lane < filled * NACT) {
        const U4 w = action_words(a.action_seed, t0 + (uint64_t)(lane / NACT), group0 + (uint64_t)(lane % NACT));
        reinterpret_cast<uint4 *>(lds_act)[lane] = make_uint4(w.x, w.y, w.z, w.w);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // per-step output bases (scalars)
    char *p_obs = reinterpret_cast<char *>(a.obs);
    char *p_rew = reinterpret_cast<char *>(a.reward);
    char *p_act = reinterpret_cast<char *>(a.actions_out);
    char *p_term = reinterpret_cast<char *>(a.terminated);
    char *p_trunc = reinterpret_cast<char *>(a.truncated);
    char *p_fin = FULL ? nullptr : reinterpret_cast<char *>(a.final_obs);
    const int64_t slice = a.slice;
    const uint32_t rew_b = rew_f32 ? 4u : 8u;
    const uint32_t act_b = (NA > 0 && !act_i32) ? 8u : 4u;

    settle_entry_loads();
    for (int step = 0; step < a.K; ++step) {
        const uint64_t t = t0 + (uint64_t)step;

        // ---- this step's actions ----
        int ai[E];
        float af[E];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < E; ++j)
            action_from_word<ENV, DEF>(P, lds_act[(step % H) * TILE + j * kWave + lane], ai[j], af[j]);
        if (FULL || p_act != nullptr) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if (!valid[j]) continue;
                char *q = p_act + le[j] * act_b;
                if constexpr (NA > 0) {
                    if (act_i32)
                        *reinterpret_cast<int32_t *>(q) = ai[j];
                    else
                        *reinterpret_cast<int64_t *>(q) = (int64_t)ai[j];
                } else {
                    *reinterpret_cast<float *>(q) = af[j];
                }
            }
        }

        // ---- dynamics + TimeLimit, E independent chains ----
        float obs[E][O];
        double rew[E];
        bool term[E], trunc[E], pend[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            term[j] = EV::template step<DEF, SAFE>(P, s[j], aux[j], el[j] == 0, ai[j], af[j], rew[j], obs[j]);
            el[j] += 1;                                              // time_limit.py:51
            trunc[j] = (a.max_steps > 0) && (el[j] >= a.max_steps);  // time_limit.py:53-54
            pend[j] = valid[j] && (term[j] || trunc[j]);
        }
        if (ep_on) {  // record_episode_statistics.py:119-143
#pragma unroll
            for (int j = 0; j < E; ++j) {
                er[j] = (float)((double)er[j] + rew[j]);  // float32 array += float64 rewards
                if (pend[j]) {
                    if (p_epr) p_epr[le[j]] = er[j];
                    if (p_epl) p_epl[le[j]] = el[j];
                    er[j] = 0.0f;
                }
            }
        }

        // ---- compact the finished envs of the wave (sync_vector_env.py:152-156) ----
        uint32_t slot[E];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint64_t m = __ballot(pend[j]);
            slot[j] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            total += (uint32_t)__popcll(m);
            if (pend[j]) {
                lds_q[slot[j]] = (uint32_t)(j * kWave + lane);
                if (!FULL && p_fin != nullptr) store_obs<O>(reinterpret_cast<float *>(p_fin), le[j], obs[j]);  // info["final_observation"]
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- at most ONE masked Philox call: this step's reset draws + action words of future steps ----
        auto draw_reset = [&](uint32_t i, const U4 &w) {
            Entry r;
            EV::reset(w, a.b0, a.b1, r.s);
            if constexpr (EV::AUX > 0) {
                EV::observe(r.s, r.o, r.x);
            } else {
                double none[1];
                EV::observe(r.s, r.o, none);
            }
            lds_res[i] = r;
        };
        auto reset_key = [&](uint32_t i) -> uint64_t {
            const uint32_t q = lds_q[i];
            const uint32_t e = (uint32_t)tile0 + q;
            return a.seeds ? landed(a.seeds[e]) : a.base_seed + a.env0 + (uint64_t)e;
        };
        // wave-uniform schedule.  Ring capacity: step's own slot was consumed above, so steps (step, step + H] fit.
        const int horizon = (a.K < step + 1 + H) ? a.K : step + 1 + H;
        const int room = horizon - filled;                               // steps that may be drawn now
        const bool must = (filled == step + 1) && (step + 1 < a.K);      // the next step has no action words yet
        const int keep = must ? NACT : 0;
        const int rlanes = (int)total < kWave - keep ? (int)total : kWave - keep;  // reset draws in this call
        int nfit = (kWave - rlanes) / NACT;                              // future steps that fit beside them
        nfit = nfit < room ? nfit : room;
        if (rlanes == 0 && !must) nfit = 0;                              // nothing forces a call: skip it
        if (rlanes > 0 || nfit > 0) {
            const bool is_act = lane < nfit * NACT;
            const int i = lane - (kWave - rlanes);                       // reset slot of the top lanes
            if (is_act || i >= 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                U4 c;
                uint64_t key;
                const int q = filled + lane / NACT;                      // launch-relative step drawn by an action lane
                if (is_act) {
                    const uint64_t g = group0 + (uint64_t)(lane % NACT), tq = t0 + (uint64_t)q;
                    c.x = (uint32_t)g; c.y = (uint32_t)(g >> 32); c.z = (uint32_t)tq;
                    c.w = ((uint32_t)(tq >> 32) & 0x0fffffffu) | (kStreamAction << 28);
                    key = a.action_seed;
                } else {
                    c.x = (uint32_t)t; c.y = (uint32_t)(t >> 32); c.z = 0u; c.w = (kStreamReset << 28);
                    key = reset_key((uint32_t)i);
                }
                const U4 w = philox4x32_10_vkey(c, (uint32_t)key, (uint32_t)(key >> 32));
                if (is_act)
                    reinterpret_cast<uint4 *>(lds_act)[(q % H) * NACT + lane % NACT] = make_uint4(w.x, w.y, w.z, w.w);
                else
                    draw_reset((uint32_t)i, w);
            }
            filled += nfit;
        }
        for (uint32_t base = (uint32_t)rlanes; base < total; base += kWave) {  // rare: more finished envs than lanes
            const uint32_t i = base + (uint32_t)lane;
            if (i < total) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                draw_reset(i, reset_words(reset_key(i), t, 0u));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- owners take the new state + observation; this step's outputs ----
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (pend[j]) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const Entry r = lds_res[slot[j]];
#pragma unroll
                for (int k = 0; k < S; ++k) s[j][k] = r.s[k];
                if constexpr (EV::AUX > 0) {
#pragma unroll
                    for (int k = 0; k < EV::AUX; ++k) aux[j][k] = r.x[k];
                }
#pragma unroll
                for (int k = 0; k < O; ++k) obs[j][k] = r.o[k];
                el[j] = 0;  // time_limit.py:67
            }
            if (!valid[j]) continue;
            store_obs<O>(reinterpret_cast<float *>(p_obs), le[j], obs[j]);
            if (FULL || p_rew != nullptr) {
                char *q = p_rew + le[j] * rew_b;
                if (rew_f32)
                    *reinterpret_cast<float *>(q) = (float)rew[j];
                else
                    *reinterpret_cast<double *>(q) = rew[j];
            }
            if (FULL || p_term != nullptr) reinterpret_cast<uint8_t *>(p_term)[le[j]] = term[j] ? 1 : 0;
            if (FULL || p_trunc != nullptr) reinterpret_cast<uint8_t *>(p_trunc)[le[j]] = trunc[j] ? 1 : 0;
        }

        // ---- advance the scalar output bases to the next trajectory slice ----
        p_obs += slice * (int64_t)(O * sizeof(float));
        if (FULL || p_rew != nullptr) p_rew += slice * (int64_t)rew_b;
        if (FULL || p_act != nullptr) p_act += slice * (int64_t)act_b;
        if (FULL || p_term != nullptr) p_term += slice;
        if (FULL || p_trunc != nullptr) p_trunc += slice;
        if (!FULL && p_fin != nullptr) p_fin += slice * (int64_t)(O * sizeof(float));
        if (!FULL && p_epr != nullptr) p_epr += slice;
        if (!FULL && p_epl != nullptr) p_epl += slice;
    }

#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (!valid[j]) continue;
#pragma unroll
        for (int k = 0; k < S; ++k) a.state[(int64_t)k * n + le[j]] = s[j][k];
        a.elapsed[le[j]] = el[j];
        if (ep_on) a.ep_acc[le[j]] = er[j];
    }
}

// ------------------------------------------------------------------------------------------------------------
// rollout_kernel_v2: same contract, tile and store pattern as rollout_kernel, with the cross-lane hand-offs removed.
// A finished env is reset by ITS OWN lane (key and counter come from registers; the new state, aux values and observation
// never leave the lane), and the lanes without a finished env draw the action words of future steps, ranked among
// themselves with mbcnt: the k-th free lane draws group k % NACT of step filled + k / NACT.  Which lane evaluates
// Philox(g, t) does not change its value, so the RNG contract and every output bit are those of rollout_kernel.
// What is left in LDS is the 1-KiB ring of action words: per wave-step ONE LDS round trip (the ring read, prefetched
// one step ahead) instead of three (ring read, compacted-list read, reset-entry read), 1 KiB instead of 9.7 KiB per
// workgroup.  A lane with two finished envs (E = 2: ~0.2 % of wave-steps) or a wave without enough free lanes for a
// forced refill (every Pendulum env truncating at step 200) takes extra passes.
// ------------------------------------------------------------------------------------------------------------
template <int ENV, bool DEF, int E, bool SAFE, int OUT = 0, int WAVES = MXV_ROLLOUT_V2_WAVES>
__global__ void __launch_bounds__(kWave * WAVES, MXV_ROLLOUT_MIN_WAVES) rollout_kernel_v2(const StepArgs a) {
    using EV = Env<ENV>;
    constexpr int S = EV::S, O = EV::O, NA = EV::NA;
    constexpr int TILE = E * kWave;
    constexpr int NACT = TILE / 4;        // lanes that draw the action words of ONE step
    constexpr int H = kWave / NACT;       // steps of action words one full call produces = depth of the ring
    static_assert(NACT <= kWave, "E must be <= 4");
    constexpr int AUXN = EV::AUX > 0 ? EV::AUX : 1;
    // WAVES waves per workgroup, each an independent tile with a private ring (no barrier anywhere): a workgroup only groups
    // WAVES consecutive tiles onto one CU so that their stores of a step land next to each other
    __shared__ uint32_t lds_ring[WAVES][H * TILE];  // ring of action words: slot (q % H) holds step q of this launch
    uint32_t *lds_act = lds_ring[threadIdx.x / kWave];

    const int lane = threadIdx.x % kWave;
    const uint32_t tile = xcd_contiguous_tile(blockIdx.x, gridDim.x) * WAVES + threadIdx.x / kWave;
    const int64_t tile0 = (int64_t)tile * TILE;
    const int64_t n = a.n;
    const Par<DEF> P(a.P);
    const uint64_t t0 = a.t + (a.t_dev ? *a.t_dev : 0);
    // OUT != 0: every per-step output array is present, no final_obs, no episode statistics, dtypes fixed (1: float64 rewards
    // + int64 actions, 2: float32 + int32) — the trajectory-recording launch.  The ~25 wave-uniform branches and ~60 scalar
    // instructions per step that the optional outputs cost fold away at compile time.
    constexpr bool FULL = OUT != 0;
    const bool act_i32 = FULL ? (OUT == 2) : ((a.flags & MXV_FLAG_ACTION_I32) != 0);
    const bool rew_f32 = FULL ? (OUT == 2) : ((a.flags & MXV_FLAG_REWARD_F32) != 0);
    const bool ep_on = !FULL && a.ep_acc != nullptr;
    const uint64_t group0 = (a.env0 + (uint64_t)tile0) >> 2;

    double s[E][S], aux[E][AUXN];
    int32_t el[E];
    bool valid[E];
    uint32_t le[E];  // env index inside the shard (fits 32 bits: mxv_create caps num_envs)
    uint64_t seed[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t e = tile0 + j * kWave + lane;
        valid[j] = e < n;
        le[j] = (uint32_t)(valid[j] ? e : 0);
#pragma unroll
        for (int k = 0; k < S; ++k) s[j][k] = a.state[(int64_t)k * n + le[j]];
        el[j] = a.elapsed[le[j]];
        seed[j] = a.seeds ? landed(a.seeds[le[j]]) : a.base_seed + a.env0 + (uint64_t)le[j];
        EV::prime(s[j], aux[j]);
    }
    float er[E];  // running episode return (RecordEpisodeStatistics.episode_returns)
#pragma unroll
    for (int j = 0; j < E; ++j) er[j] = ep_on ? a.ep_acc[le[j]] : 0.0f;
    float *p_epr = FULL ? nullptr : a.ep_return_out;
    int32_t *p_epl = FULL ? nullptr : a.ep_length_out;

    // action words of the first min(H, K) steps: lane L draws group L % NACT of step L / NACT
    int filled = a.K < H ? a.K : H;  // steps [0, filled) of this launch have their action words in the ring
    if (lane < filled * NACT) {
        const U4 w = action_words(a.action_seed, t0 + (uint64_t)(lane / NACT), group0 + (uint64_t)(lane % NACT));
        reinterpret_cast<uint4 *>(lds_act)[lane] = make_uint4(w.x, w.y, w.z, w.w);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t word[E];  // this step's action words (read one step ahead)
#pragma unroll
    for (int j = 0; j < E; ++j) word[j] = lds_act[j * kWave + lane];

    char *p_obs = reinterpret_cast<char *>(a.obs);
    char *p_rew = reinterpret_cast<char *>(a.reward);
    char *p_act = reinterpret_cast<char *>(a.actions_out);
    char *p_term = reinterpret_cast<char *>(a.terminated);
    char *p_trunc = reinterpret_cast<char *>(a.truncated);
    char *p_fin = FULL ? nullptr : reinterpret_cast<char *>(a.final_obs);
    const int64_t slice = a.slice;
    const uint32_t rew_b = rew_f32 ? 4u : 8u;
    const uint32_t act_b = (NA > 0 && !act_i32) ? 8u : 4u;

    settle_entry_loads();
    for (int step = 0; step < a.K; ++step) {
        const uint64_t t = t0 + (uint64_t)step;

        // ---- this step's actions ----
        int ai[E];
        float af[E];
#pragma unroll
        for (int j = 0; j < E; ++j) action_from_word<ENV, DEF>(P, word[j], ai[j], af[j]);
        if (FULL || p_act != nullptr) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                if (!valid[j]) continue;
                char *q = p_act + le[j] * act_b;
                if constexpr (NA > 0) {
                    if (act_i32)
                        *reinterpret_cast<int32_t *>(q) = ai[j];
                    else
                        *reinterpret_cast<int64_t *>(q) = (int64_t)ai[j];
                } else {
                    *reinterpret_cast<float *>(q) = af[j];
                }
            }
        }

        // ---- dynamics + TimeLimit, E independent chains ----
        float obs[E][O];
        double rew[E];
        bool term[E], trunc[E], pend[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            term[j] = EV::template step<DEF, SAFE>(P, s[j], aux[j], el[j] == 0, ai[j], af[j], rew[j], obs[j]);
            el[j] += 1;                                              // time_limit.py:51
            trunc[j] = (a.max_steps > 0) && (el[j] >= a.max_steps);  // time_limit.py:53-54
            pend[j] = valid[j] && (term[j] || trunc[j]);
        }
        if (ep_on) {  // record_episode_statistics.py:119-143
#pragma unroll
            for (int j = 0; j < E; ++j) {
                er[j] = (float)((double)er[j] + rew[j]);  // float32 array += float64 rewards
                if (pend[j]) {
                    if (p_epr) p_epr[le[j]] = er[j];
                    if (p_epl) p_epl[le[j]] = el[j];
                    er[j] = 0.0f;
                }
            }
        }
        // outputs that do not depend on the reset go out first: reward, flags, info["final_observation"]
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (!valid[j]) continue;
            if (FULL || p_rew != nullptr) {
                char *q = p_rew + le[j] * rew_b;
                if (rew_f32)
                    *reinterpret_cast<float *>(q) = (float)rew[j];
                else
                    *reinterpret_cast<double *>(q) = rew[j];
            }
            if (FULL || p_term != nullptr) reinterpret_cast<uint8_t *>(p_term)[le[j]] = term[j] ? 1 : 0;
            if (FULL || p_trunc != nullptr) reinterpret_cast<uint8_t *>(p_trunc)[le[j]] = trunc[j] ? 1 : 0;
            if (!FULL && pend[j] && p_fin != nullptr) store_obs<O>(reinterpret_cast<float *>(p_fin), le[j], obs[j]);
        }

        // ---- masked Philox passes: every lane with a finished env resets it (sync_vector_env.py:152-156); the free
        //      lanes draw action words of future steps.  Normally exactly one pass, or none at all. ----
        while (true) {
            int jsel = -1;
#pragma unroll
            for (int j = E - 1; j >= 0; --j)
                if (pend[j]) jsel = j;
            const bool resets = jsel >= 0;
            const uint64_t busy = __ballot(resets);
            const bool must = (filled == step + 1) && (step + 1 < a.K);      // the next step has no action words yet
            if (busy == 0 && !must) break;
            // ring capacity: this step's slot is consumed, so steps (step, step + H] fit
            const int horizon = (a.K < step + 1 + H) ? a.K : step + 1 + H;
            const int room = horizon - filled;
            const int nfree = kWave - (int)__popcll(busy);
            int nfit = nfree / NACT;                                         // future steps the free lanes can draw
            nfit = nfit < room ? nfit : room;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(~busy >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)~busy, 0u));
            const bool is_act = !resets && (int)rank < nfit * NACT;
            if (resets || is_act) {
                U4 c;
                uint64_t key;
                const int q = filled + (int)rank / NACT;                     // launch-relative step drawn by an action lane
                if (is_act) {
                    const uint64_t g = group0 + (uint64_t)(rank % NACT), tq = t0 + (uint64_t)q;
                    c.x = (uint32_t)g; c.y = (uint32_t)(g >> 32); c.z = (uint32_t)tq;
                    c.w = ((uint32_t)(tq >> 32) & 0x0fffffffu) | (kStreamAction << 28);
                    key = a.action_seed;
                } else {
                    c.x = (uint32_t)t; c.y = (uint32_t)(t >> 32); c.z = 0u; c.w = (kStreamReset << 28);
                    key = seed[0];
#pragma unroll
                    for (int j = 1; j < E; ++j) key = (j == jsel) ? seed[j] : key;
                }
                const U4 w = philox4x32_10_vkey(c, (uint32_t)key, (uint32_t)(key >> 32));
                if (is_act) {
                    reinterpret_cast<uint4 *>(lds_act)[(q % H) * NACT + rank % NACT] = make_uint4(w.x, w.y, w.z, w.w);
                } else {
                    double ns[S], naux[AUXN];
                    float nobs[O];
                    EV::reset(w, a.b0, a.b1, ns);
                    EV::observe(ns, nobs, naux);
#pragma unroll
                    for (int j = 0; j < E; ++j)
                        if (j == jsel) {
#pragma unroll
                            for (int k = 0; k < EV::AUX; ++k) aux[j][k] = naux[k];
#pragma unroll
                            for (int k = 0; k < S; ++k) s[j][k] = ns[k];
#pragma unroll
                            for (int k = 0; k < O; ++k) obs[j][k] = nobs[k];
                            el[j] = 0;  // time_limit.py:67
                            pend[j] = false;
                        }
                }
            }
            filled += nfit;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        // ---- next step's action words (LDS latency hides behind the observation stores) ----
        if (step + 1 < a.K) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < E; ++j) word[j] = lds_act[((step + 1) % H) * TILE + j * kWave + lane];
        }
#pragma unroll
        for (int j = 0; j < E; ++j)
            if (valid[j]) store_obs<O>(reinterpret_cast<float *>(p_obs), le[j], obs[j]);

        // ---- advance the scalar output bases to the next trajectory slice ----
        p_obs += slice * (int64_t)(O * sizeof(float));
        if (FULL || p_rew != nullptr) p_rew += slice * (int64_t)rew_b;
        if (FULL || p_act != nullptr) p_act += slice * (int64_t)act_b;
        if (FULL || p_term != nullptr) p_term += slice;
        if (FULL || p_trunc != nullptr) p_trunc += slice;
        if (!FULL && p_fin != nullptr) p_fin += slice * (int64_t)(O * sizeof(float));
        if (!FULL && p_epr != nullptr) p_epr += slice;
        if (!FULL && p_epl != nullptr) p_epl += slice;
    }

#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (!valid[j]) continue;
#pragma unroll
        for (int k = 0; k < S; ++k) a.state[(int64_t)k * n + le[j]] = s[j][k];
        a.elapsed[le[j]] = el[j];
        if (ep_on) a.ep_acc[le[j]] = er[j];
    }
}

// Explicit reset (SyncVectorEnv.reset_wait, sync_vector_env.py:90-129): one env per lane.
template <int ENV>
__global__ void __launch_bounds__(kBlock) reset_kernel(const ResetArgs a) {
    using EV = Env<ENV>;
    constexpr int S = EV::S, O = EV::O;
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= a.n) return;
    if (a.mask != nullptr && a.mask[e] == 0) return;
    const uint64_t seed = a.seeds ? landed(a.seeds[e]) : a.base_seed + a.env0 + (uint64_t)e;
    const U4 w = reset_words(seed, a.t, a.r);
    double s[S];
    EV::reset(w, a.b0, a.b1, s);
#pragma unroll
    for (int k = 0; k < S; ++k) a.state[(int64_t)k * a.n + e] = s[k];
    a.elapsed[e] = 0;  // time_limit.py:67
    if (a.ep_acc != nullptr) a.ep_acc[e] = 0.0f;  // record_episode_statistics.py:91-94
    if (a.obs != nullptr) {
        float o[O];
        double aux_unused[EV::AUX > 0 ? EV::AUX : 1];
        EV::observe(s, o, aux_unused);
        store_obs<O>(a.obs, e, o);
    }
}

// action_space.sample() without stepping: one Philox call (4 envs) per lane.
template <int ENV, int DEF>
__global__ void __launch_bounds__(kBlock) sample_kernel(const SampleArgs a) {
    constexpr int NA = Env<ENV>::NA;
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // local group index
    if (c * 4 >= a.n) return;
    const Par<DEF> P(a.P, a.params_pe, a.n);
    const uint64_t t = a.t + (a.t_dev ? *a.t_dev : 0);
    const U4 w = action_words(a.action_seed, t, (a.env0 >> 2) + (uint64_t)c);
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t e = c * 4 + q;
        if (e >= a.n) break;
        int ai;
        float af;
        action_from_word<ENV, DEF>(P.at(e), ws[q], ai, af);
        if constexpr (NA > 0) {
            if (a.flags & MXV_FLAG_ACTION_I32)
                static_cast<int32_t *>(a.actions_out)[e] = ai;
            else
                static_cast<int64_t *>(a.actions_out)[e] = ai;
        } else {
            static_cast<float *>(a.actions_out)[e] = af;
        }
    }
}

__global__ void set_word_kernel(uint64_t *dst, uint64_t value) { *dst = value; }

// rollout_kernel (resets compacted through LDS) or rollout_kernel_v2 (owner-lane resets, prefetched action words), per env
// kind from the same-box A/B in profiles/r01h_rollout_v2_ab.txt: v2 wins 2-4 % for Pendulum / MountainCar /
// MountainCarContinuous, v1 1-2 % for CartPole and Acrobot.  MXV_ROLLOUT_V2 = 0 / 1 forces one of them (tuning builds).
template <int ENV>
constexpr bool use_rollout_v2() {
#if MXV_ROLLOUT_V2 == 0
    return false;
#elif MXV_ROLLOUT_V2 == 1
    return true;
#else
    return ENV == MXV_PENDULUM || ENV == MXV_MOUNTAINCAR || ENV == MXV_MOUNTAINCAR_CONT;
#endif
}
template <int ENV, bool DEF, int ER, bool SAFE, int OUT>
void launch_rollout_out(unsigned grid, hipStream_t stream, const StepArgs &a) {
    if constexpr (use_rollout_v2<ENV>())
        hipLaunchKernelGGL((rollout_kernel_v2<ENV, DEF, ER, SAFE, OUT>), dim3((grid + MXV_ROLLOUT_V2_WAVES - 1) / MXV_ROLLOUT_V2_WAVES),
                           dim3(kWave * MXV_ROLLOUT_V2_WAVES), 0, stream, a);
    else
        hipLaunchKernelGGL((rollout_kernel<ENV, DEF, ER, SAFE, OUT>), dim3(grid), dim3(kWave), 0, stream, a);
}
template <int ENV, bool DEF, int ER, bool SAFE>
void launch_rollout(unsigned grid, hipStream_t stream, const StepArgs &a) {
    // the trajectory-recording shape (all outputs, no final_obs / statistics) has its own straight-line instantiations
    int out = 0;
    if (DEF && a.reward && a.actions_out && a.terminated && a.truncated && !a.final_obs && !a.ep_acc) {
        const int f = a.flags & (MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32);
        out = f == 0 ? 1 : (f == (MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32) ? 2 : 0);
    }
    if constexpr (DEF) {
        if (out == 1) return launch_rollout_out<ENV, DEF, ER, SAFE, 1>(grid, stream, a);
        if (out == 2) return launch_rollout_out<ENV, DEF, ER, SAFE, 2>(grid, stream, a);
    }
    launch_rollout_out<ENV, DEF, ER, SAFE, 0>(grid, stream, a);
}

template <int ENV>
hipError_t launch_step_env(int pm, const StepArgs &a, hipStream_t stream) {
    const bool def = pm == PM_DEFAULT;
    // Sampled actions + autoreset, several steps per launch: the fused fast path.  (Single-step launches stay on
    // step_kernel: they are latency-bound and its 58 VGPRs give twice the occupancy.)
    if (a.actions == nullptr && !(a.flags & MXV_FLAG_NO_AUTORESET) && a.K > 1 && pm != PM_PER_ENV && !a.step_noise) {
        constexpr int ER = rollout_envs_per_lane(ENV);
        const int64_t rtile = (int64_t)ER * kWave;
        const unsigned rgrid = (unsigned)((a.n + rtile - 1) / rtile);
        bool fast = false;
        if constexpr (ENV == MXV_CARTPOLE) fast = def && !a.state_injected;  // see Env<MXV_CARTPOLE>::step, SAFE
        if (!def) {
            launch_rollout<ENV, false, ER, true>(rgrid, stream, a);
        } else if (fast) {
            if constexpr (ENV == MXV_CARTPOLE) launch_rollout<ENV, true, ER, false>(rgrid, stream, a);
        } else {
            launch_rollout<ENV, true, ER, true>(rgrid, stream, a);
        }
        return hipGetLastError();
    }
    constexpr int E = envs_per_lane(ENV);
    constexpr bool C = MXV_CONSEC != 0;
    const int64_t tile = (int64_t)E * kBlock;
    const unsigned grid = (unsigned)((a.n + tile - 1) / tile);
    if (a.K > 1) {
        if (pm == PM_DEFAULT)
            hipLaunchKernelGGL((step_kernel<ENV, PM_DEFAULT, E, C, true>), dim3(grid), dim3(kBlock), 0, stream, a);
        else if (pm == PM_BROADCAST)
            hipLaunchKernelGGL((step_kernel<ENV, PM_BROADCAST, E, C, true>), dim3(grid), dim3(kBlock), 0, stream, a);
        else
            hipLaunchKernelGGL((step_kernel<ENV, PM_PER_ENV, E, C, true>), dim3(grid), dim3(kBlock), 0, stream, a);
    } else {
        if (pm == PM_DEFAULT)
            hipLaunchKernelGGL((step_kernel<ENV, PM_DEFAULT, E, C, false>), dim3(grid), dim3(kBlock), 0, stream, a);
        else if (pm == PM_BROADCAST)
            hipLaunchKernelGGL((step_kernel<ENV, PM_BROADCAST, E, C, false>), dim3(grid), dim3(kBlock), 0, stream, a);
        else
            hipLaunchKernelGGL((step_kernel<ENV, PM_PER_ENV, E, C, false>), dim3(grid), dim3(kBlock), 0, stream, a);
    }
    return hipGetLastError();
}

template <int ENV>
hipError_t launch_sample_env(int pm, const SampleArgs &a, hipStream_t stream) {
    const int64_t groups = (a.n + 3) / 4;
    const unsigned grid = (unsigned)((groups + kBlock - 1) / kBlock);
    if (pm == PM_DEFAULT)
        hipLaunchKernelGGL((sample_kernel<ENV, PM_DEFAULT>), dim3(grid), dim3(kBlock), 0, stream, a);
    else if (pm == PM_BROADCAST)
        hipLaunchKernelGGL((sample_kernel<ENV, PM_BROADCAST>), dim3(grid), dim3(kBlock), 0, stream, a);
    else
        hipLaunchKernelGGL((sample_kernel<ENV, PM_PER_ENV>), dim3(grid), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_step(int env_id, int default_params, const StepArgs &a, hipStream_t stream) {
    switch (env_id) {
        case MXV_CARTPOLE: return launch_step_env<MXV_CARTPOLE>(default_params, a, stream);
        case MXV_PENDULUM: return launch_step_env<MXV_PENDULUM>(default_params, a, stream);
        case MXV_ACROBOT: return launch_step_env<MXV_ACROBOT>(default_params, a, stream);
        case MXV_MOUNTAINCAR: return launch_step_env<MXV_MOUNTAINCAR>(default_params, a, stream);
        case MXV_MOUNTAINCAR_CONT: return launch_step_env<MXV_MOUNTAINCAR_CONT>(default_params, a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reset(int env_id, const ResetArgs &a, hipStream_t stream) {
    const unsigned grid = (unsigned)((a.n + kBlock - 1) / kBlock);
    switch (env_id) {
        case MXV_CARTPOLE: hipLaunchKernelGGL(reset_kernel<MXV_CARTPOLE>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case MXV_PENDULUM: hipLaunchKernelGGL(reset_kernel<MXV_PENDULUM>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case MXV_ACROBOT: hipLaunchKernelGGL(reset_kernel<MXV_ACROBOT>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case MXV_MOUNTAINCAR: hipLaunchKernelGGL(reset_kernel<MXV_MOUNTAINCAR>, dim3(grid), dim3(kBlock), 0, stream, a); break;
        case MXV_MOUNTAINCAR_CONT:
            hipLaunchKernelGGL(reset_kernel<MXV_MOUNTAINCAR_CONT>, dim3(grid), dim3(kBlock), 0, stream, a);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_sample(int env_id, int default_params, const SampleArgs &a, hipStream_t stream) {
    switch (env_id) {
        case MXV_CARTPOLE: return launch_sample_env<MXV_CARTPOLE>(default_params, a, stream);
        case MXV_PENDULUM: return launch_sample_env<MXV_PENDULUM>(default_params, a, stream);
        case MXV_ACROBOT: return launch_sample_env<MXV_ACROBOT>(default_params, a, stream);
        case MXV_MOUNTAINCAR: return launch_sample_env<MXV_MOUNTAINCAR>(default_params, a, stream);
        case MXV_MOUNTAINCAR_CONT: return launch_sample_env<MXV_MOUNTAINCAR_CONT>(default_params, a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_set_word(uint64_t *dst, uint64_t value, hipStream_t stream) {
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, stream, dst, value);
    return hipGetLastError();
}

}  // namespace mxv
