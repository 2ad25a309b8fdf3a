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

// Word `idx` (0..3, runtime) of a Philox result.
__device__ __forceinline__ uint32_t pick_word(const U4 &w, uint32_t idx) {
    return idx == 0 ? w.x : (idx == 1 ? w.y : (idx == 2 ? w.z : w.w));
}

// step_kernel<ENV, DEF, E, CONSEC>: a.K vector steps in ONE launch, env state held in registers between
// steps (K = 1 is the plain step() call).  Per step the only HBM traffic is the step's outputs; state
// and elapsed[] are read once at entry and written once at exit, i.e. 16*S/K + 8/K bytes per env-step.
//   CONSEC = false: lane owns envs tile0 + j*256 + tid (every access of a wave is a dense burst);
//                   Philox action words are transposed through LDS (one call = 4 consecutive envs).
//   CONSEC = true : lane owns the E consecutive envs tile0 + tid*E + j: a Philox action group is
//                   lane-private (no LDS, no barrier) and the flag bytes of a lane are contiguous.
template <int ENV, bool DEF, int E, bool CONSEC, bool MULTI>
__global__ void __launch_bounds__(kBlock, MXV_MIN_WAVES) step_kernel(const StepArgs a) {
    using EV = Env<ENV>;
    constexpr int S = EV::S, O = EV::O, NA = EV::NA;
    constexpr int TILE = E * kBlock;
    static_assert(!CONSEC || E == 1 || E == 2 || E % 4 == 0, "CONSEC needs E in {1, 2, 4k}");
    const int tid = threadIdx.x;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const int64_t n = a.n;
    const Par<DEF> P(a.P);
    const uint64_t t0 = a.t + (a.t_dev ? *a.t_dev : 0);
    const bool autoreset = !(a.flags & MXV_FLAG_NO_AUTORESET);
    const bool sampled = a.actions == nullptr;
    auto env_of = [&](int j) -> int64_t { return CONSEC ? tile0 + (int64_t)tid * E + j : tile0 + (int64_t)j * kBlock + tid; };

    // ---- entry: state + elapsed of the lane's E envs ----
    double s[E][S];
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
    }
    __shared__ uint32_t sw[CONSEC ? 4 : TILE];

    const int nsteps = MULTI ? a.K : 1;  // MULTI = false: the plain step() launch, no loop-carried bookkeeping
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
                        action_from_word<ENV, DEF>(P, pick_word(w, (uint32_t)((ge + q) & 3)), ai[j0 + q], af[j0 + q]);
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
                for (int j = 0; j < E; ++j) action_from_word<ENV, DEF>(P, sw[j * kBlock + tid], ai[j], af[j]);
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

        // ---- dynamics + TimeLimit, E independent chains ----
        float obs[E][O];
        double rew[E];
        bool term[E], trunc[E], pend[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            term[j] = EV::template step<DEF>(P, s[j], el[j] == 0, ai[j], af[j], rew[j], obs[j]);
            el[j] += 1;                                              // time_limit.py:51
            trunc[j] = (a.max_steps > 0) && (el[j] >= a.max_steps);  // time_limit.py:53-54
            pend[j] = autoreset && (term[j] || trunc[j]);
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
                const uint64_t seed = a.seeds ? a.seeds[v ? e : 0] : a.base_seed + a.env0 + (uint64_t)e;
                const U4 w = reset_words(seed, t, 0u);
                double ns[S];
                float nobs[O];
                EV::reset(w, a.b0, a.b1, ns);
                EV::observe(ns, nobs);
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (j == jsel) {
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
    const uint64_t seed = a.seeds ? a.seeds[e] : a.base_seed + a.env0 + (uint64_t)e;
    const U4 w = reset_words(seed, a.t, a.r);
    double s[S];
    EV::reset(w, a.b0, a.b1, s);
#pragma unroll
    for (int k = 0; k < S; ++k) a.state[(int64_t)k * a.n + e] = s[k];
    a.elapsed[e] = 0;  // time_limit.py:67
    if (a.obs != nullptr) {
        float o[O];
        EV::observe(s, o);
        store_obs<O>(a.obs, e, o);
    }
}

// action_space.sample() without stepping: one Philox call (4 envs) per lane.
template <int ENV, bool DEF>
__global__ void __launch_bounds__(kBlock) sample_kernel(const SampleArgs a) {
    constexpr int NA = Env<ENV>::NA;
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // local group index
    if (c * 4 >= a.n) return;
    const Par<DEF> P(a.P);
    const uint64_t t = a.t + (a.t_dev ? *a.t_dev : 0);
    const U4 w = action_words(a.action_seed, t, (a.env0 >> 2) + (uint64_t)c);
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t e = c * 4 + q;
        if (e >= a.n) break;
        int ai;
        float af;
        action_from_word<ENV, DEF>(P, ws[q], ai, af);
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

template <int ENV>
hipError_t launch_step_env(bool def, const StepArgs &a, hipStream_t stream) {
    constexpr int E = envs_per_lane(ENV);
    constexpr bool C = MXV_CONSEC != 0;
    const int64_t tile = (int64_t)E * kBlock;
    const unsigned grid = (unsigned)((a.n + tile - 1) / tile);
    if (a.K > 1) {
        if (def)
            hipLaunchKernelGGL((step_kernel<ENV, true, E, C, true>), dim3(grid), dim3(kBlock), 0, stream, a);
        else
            hipLaunchKernelGGL((step_kernel<ENV, false, E, C, true>), dim3(grid), dim3(kBlock), 0, stream, a);
    } else {
        if (def)
            hipLaunchKernelGGL((step_kernel<ENV, true, E, C, false>), dim3(grid), dim3(kBlock), 0, stream, a);
        else
            hipLaunchKernelGGL((step_kernel<ENV, false, E, C, false>), dim3(grid), dim3(kBlock), 0, stream, a);
    }
    return hipGetLastError();
}

template <int ENV>
hipError_t launch_sample_env(bool def, const SampleArgs &a, hipStream_t stream) {
    const int64_t groups = (a.n + 3) / 4;
    const unsigned grid = (unsigned)((groups + kBlock - 1) / kBlock);
    if (def)
        hipLaunchKernelGGL((sample_kernel<ENV, true>), dim3(grid), dim3(kBlock), 0, stream, a);
    else
        hipLaunchKernelGGL((sample_kernel<ENV, false>), dim3(grid), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_step(int env_id, bool default_params, const StepArgs &a, hipStream_t stream) {
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

hipError_t launch_sample(int env_id, bool default_params, const SampleArgs &a, hipStream_t stream) {
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
