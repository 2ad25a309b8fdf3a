// mxv_kernels.hpp — launch interface between the C-ABI host code (mxv_api.cpp) and the
// gfx950 kernels (mxv_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mxv_device.hpp"

namespace mxv {

struct StepArgs {
    double *state;         // [S][N] fp64, struct-of-arrays
    void *elapsed;         // [N] TimeLimit counters: uint16 when elapsed16 (every handle whose limit fits: 2 B instead of 4 each way per step), else int32
    uint32_t *episodes;    // [N] resets each env has had since seeding = index of its next draw from the reset stream
    const void *actions;   // int64/int32/float32 [N]; nullptr -> draw from the Philox action stream
    void *actions_out;     // optional record of the sampled actions
    float *obs;            // [N][O]
    void *reward;          // double[N] (float[N] with MXV_FLAG_REWARD_F32); may be nullptr
    uint8_t *terminated;   // [N]; may be nullptr
    uint8_t *truncated;    // [N]; may be nullptr
    float *final_obs;      // [N][O], rows of finished envs only; may be nullptr
    const uint64_t *seeds; // per-env seeds or nullptr (base_seed + global index)
    const uint64_t *t_dev; // optional device-resident base step index (hipGraph replay)
    double *ret_state;     // STATS bit 1: [N] NormalizeReward's running discounted returns (read at entry, written at exit), with
    double *ret_part;      //   [K][tiles][2] their per-tile sum / sum of squares after every step's update, and
    double ret_gamma;      //   the discount
    double *obs_part;      // STATS launches: [K][tiles][2 O] column sums / sums of squares of every step's observations, or nullptr
    uint64_t *clock_out;   // device clock advanced by the launch itself (step_kernel, small grids): = t_dev, with clock_ticket; or nullptr
    uint32_t *clock_ticket; // workgroups that have finished (zero between launches)
    int32_t *err;          // latched error word (bit 0: invalid discrete action)
    int64_t n;             // local envs
    uint64_t env0;         // global index of local env 0
    uint64_t base_seed;
    uint64_t action_seed;
    uint64_t t;            // step index (added to *t_dev when t_dev != nullptr)
    double b0, b1;         // reset bounds
    int32_t max_steps;     // TimeLimit, <= 0 disables
    int32_t flags;         // MXV_FLAG_*
    int32_t K;             // vector steps fused into this launch (>= 1)
    int32_t state_injected; // mxv_set_state() ran since the last launch: no invariant on the state may be assumed
    int32_t elapsed16;      // storage type of elapsed[] (see above)
    const float *hi_in;     // HILO launches (mxv_adopt_obs): the observation rows [N][O] the previous step / split left = the float32 half of the state
    int32_t *lo;            // HILO launches: [N][S] (row-major, like the observations) the other half, (state - (double)float32(state)) in units of 2^(exponent(hi) - 53); nullptr otherwise
    int32_t step_noise;     // Acrobot torque_noise_max > 0 somewhere: draw the step-noise word (step_kernel launches only)
    int64_t slice;         // output pointers advance by `slice` envs per step ([K][N] trajectories) or 0
    int64_t act_slice;     // action tape advance per step (envs) or 0
    const double *params_pe; // [MXV_MAX_PARAMS][N] per-env physics parameters (PM_PER_ENV launches) or nullptr
    uint8_t *beyond;         // CartPole without autoreset: [N] "this env has terminated before" (cartpole.py:169-184) or nullptr
    // episode statistics (gym/wrappers/record_episode_statistics.py:96-151), all nullptr when disabled
    float *ep_acc;           // [N] running episode return, float32 like the reference's accumulator
    float *ep_return_out;    // [N] / [K][N]: episode return, written only where terminated | truncated
    int32_t *ep_length_out;  // [N] / [K][N]: episode length, written only where terminated | truncated
    // optional second destination of the LAST step's outputs of a fused rollout (mxv_set_final_snapshot), [N] each
    float *snap_obs;
    void *snap_reward;
    uint8_t *snap_terminated;
    uint8_t *snap_truncated;
    EnvParams P;
};

// One launch over several homogeneous segments (mixed_rollout_kernel): segment i owns workgroups [first_block[i], first_block[i+1]).
struct MixedArgs {
    StepArgs seg[MXV_MAX_MIXED];
    uint32_t first_block[MXV_MAX_MIXED + 1];
    int32_t kind[MXV_MAX_MIXED];   // env kind of each segment
    int32_t count;
};

struct ResetArgs {
    double *state;
    void *elapsed;         // as in StepArgs
    int32_t elapsed16;
    uint32_t *episodes;    // [N] reset ordinals (read: index of this draw; written back + 1)
    float *obs;            // may be nullptr
    const uint8_t *mask;   // may be nullptr (all)
    const uint64_t *seeds; // may be nullptr
    float *ep_acc;         // may be nullptr: running episode returns, zeroed for the envs being reset
    uint8_t *beyond;       // may be nullptr: CartPole's steps_beyond_terminated marks, cleared for the envs being reset
    int64_t n;
    uint64_t env0;
    uint64_t base_seed;
    double b0, b1;
};

struct CompactArgs {
    const uint8_t *terminated, *truncated;  // [N]
    const float *final_obs;                 // [N][O] dense (rows of finished envs valid)
    int32_t *count;                         // total, written by the launch
    int32_t *chunk_counts;                  // [compact_chunks(N)] scratch
    int32_t *idx;                           // [N] packed env indices
    float *rows;                            // [N][O] packed rows
    const float *ep_return_in;              // [N] dense episode statistics of this step (valid at finished envs) or nullptr
    const int32_t *ep_length_in;
    float *ep_return;                       // [N] packed
    int32_t *ep_length;
    int64_t n;
};

struct SampleArgs {
    void *actions_out;
    const uint64_t *t_dev;
    int64_t n;
    uint64_t env0;
    uint64_t action_seed;
    uint64_t t;
    int32_t flags;
    const double *params_pe;
    EnvParams P;
};

// Envs per lane of the step kernel (tile = E * 256 envs per workgroup), chosen per env kind from the sweep in
// profiles/r1/r01_variant_sweep.md: two interleaved chains for the light envs (ILP without register spills; four
// chains spill SGPRs/VGPRs inside the fused loop), one chain for Pendulum and for Acrobot's RK4 (~60 live fp64).
#ifndef MXV_ENVS_PER_LANE
#define MXV_ENVS_PER_LANE 2
#endif
#ifndef MXV_ENVS_PER_LANE_PENDULUM
#define MXV_ENVS_PER_LANE_PENDULUM 1
#endif
#ifndef MXV_ENVS_PER_LANE_ACROBOT
#define MXV_ENVS_PER_LANE_ACROBOT 1
#endif
constexpr int envs_per_lane(int env_id) {
    return env_id == MXV_ACROBOT ? MXV_ENVS_PER_LANE_ACROBOT
                                 : (env_id == MXV_PENDULUM ? MXV_ENVS_PER_LANE_PENDULUM : MXV_ENVS_PER_LANE);
}
// Envs per lane of rollout_kernel (one wave64 per workgroup, tile = E * 64 envs; E <= 3).
#ifndef MXV_ROLLOUT_E
#define MXV_ROLLOUT_E 2
#endif
#ifndef MXV_ROLLOUT_E_PENDULUM
#define MXV_ROLLOUT_E_PENDULUM 1
#endif
#ifndef MXV_ROLLOUT_E_ACROBOT
#define MXV_ROLLOUT_E_ACROBOT 1
#endif
#ifndef MXV_ROLLOUT_E_MOUNTAINCAR
#define MXV_ROLLOUT_E_MOUNTAINCAR MXV_ROLLOUT_E
#endif
#ifndef MXV_ROLLOUT_E_MOUNTAINCAR_CONT
#define MXV_ROLLOUT_E_MOUNTAINCAR_CONT MXV_ROLLOUT_E
#endif
constexpr int rollout_envs_per_lane(int env_id) {
    return env_id == MXV_ACROBOT ? MXV_ROLLOUT_E_ACROBOT
           : env_id == MXV_PENDULUM ? MXV_ROLLOUT_E_PENDULUM
           : env_id == MXV_MOUNTAINCAR ? MXV_ROLLOUT_E_MOUNTAINCAR
           : env_id == MXV_MOUNTAINCAR_CONT ? MXV_ROLLOUT_E_MOUNTAINCAR_CONT
                                            : MXV_ROLLOUT_E;
}
// 1: a lane owns E consecutive envs (lane-private Philox action group); 0: wave-dense striding + LDS exchange.
#ifndef MXV_CONSEC
#define MXV_CONSEC 0
#endif
// __launch_bounds__ second argument = minimum waves per SIMD (4 => at most 128 VGPRs): a 2^20-env launch at 4 envs
// per lane is exactly 4 waves per SIMD, all of which must be co-resident to run in one round.
#ifndef MXV_MIN_WAVES
#define MXV_MIN_WAVES 1
#endif
// 1: shards smaller than one E-env-per-lane wave per SIMD run the fused rollout with one env per lane (A/B hook)
#ifndef MXV_ROLLOUT_SMALL_E1
#define MXV_ROLLOUT_SMALL_E1 1
#endif
// the shard size below which a two-envs-per-lane kind runs one env per lane: FACTOR x (one E = 2 wave per SIMD); INCLUSIVE = 1 includes
// the boundary itself, i.e. the 2^17-env shard of an 8-GPU strong-scaling job (round 3: 1.01 -> 0.92 us per CartPole step, MountainCar
// 0.85 -> 0.75, MountainCarContinuous 0.99 -> 0.79; at 2^18 two envs per lane win, 1.43 vs 1.53, at 2^19 the two are equal.  profiles/r3/r3k_small_shard_e1_ab.jsonl, r3s_e1_factor_ab.jsonl)
#ifndef MXV_ROLLOUT_E1_FACTOR
#define MXV_ROLLOUT_E1_FACTOR 1
#endif
#ifndef MXV_ROLLOUT_E1_INCLUSIVE
#define MXV_ROLLOUT_E1_INCLUSIVE 1
#endif
// steps between two look-ahead passes of rollout_kernel_v3 over the same env slot (power of two, >= envs per lane)
#ifndef MXV_ROLLOUT_PASS_PERIOD
#define MXV_ROLLOUT_PASS_PERIOD 8
#endif
// rollout_kernel's minimum waves per SIMD (tuning hook; 1 = let the register allocator decide: 113 VGPRs = 4 waves for CartPole)
#ifndef MXV_ROLLOUT_MIN_WAVES
#define MXV_ROLLOUT_MIN_WAVES 1
#endif
// 1: the fused rollout's stores take a wave-uniform scalar base + a pinned 32-bit lane offset (pin32 in mxv_kernels.hip); 0: round 2's
// 64-bit vector addresses (A/B hook)
#ifndef MXV_SADDR_STORES
#define MXV_SADDR_STORES 1
#endif
// A/B hook: 0 compiles CartPole's steps_beyond_terminated bookkeeping out of step_kernel
#ifndef MXV_CARTPOLE_BEYOND
#define MXV_CARTPOLE_BEYOND 1
#endif
// 1: XCD-aware workgroup -> tile map (see xcd_contiguous_tile in mxv_kernels.hip); 0: tiles in workgroup-id order.
#ifndef MXV_XCD_MAP
#define MXV_XCD_MAP 1
#endif
// > 0: the XCD-aware map hands each XCD blocks of this many tiles in turn instead of one contiguous eighth (tuning hook)
#ifndef MXV_XCD_BLOCK
#define MXV_XCD_BLOCK 0
#endif
constexpr int kBlock = 256;

// Which kernel instantiation a step launch took (mxv_last_launch, include/mxv.h: mxv_launch_info has the same members)
struct LaunchInfo {
    int32_t kernel;         // 0 = step_kernel, 1 = rollout_kernel_v3
    int32_t env_id;
    int32_t param_mode;     // PM_DEFAULT / PM_BROADCAST / PM_PER_ENV
    int32_t envs_per_lane;
    int32_t safe;           // guarded trigonometry (state injected / unusual bounds / non-default parameters)
    int32_t out_mode;       // rollout_kernel_v3: 0 generic body, 1 trajectory outputs float64 + int64, 2 float32 + int32
    int32_t tape;           // actions from a caller's tape
    int32_t steps;          // K
    uint32_t grid, block;
};
// param_mode: PM_DEFAULT / PM_BROADCAST / PM_PER_ENV (mxv_device.hpp)
hipError_t launch_step(int env_id, int param_mode, const StepArgs &a, hipStream_t stream, LaunchInfo *info = nullptr);
// The fp64 state as (float32 observation, int32 residual) pairs — see hilo_encode in mxv_kernels.hip.  Kinds whose observation IS float32(state):
bool hilo_supported(int env_id);
hipError_t launch_hilo_step(int env_id, const StepArgs &a, hipStream_t stream, LaunchInfo *info = nullptr);   // one step, default parameters, a.hi_in / a.lo set
hipError_t launch_hilo_split(int env_id, const double *state, float *hi_obs, int32_t *lo, double *side, int64_t n, hipStream_t stream);
hipError_t launch_hilo_join(int env_id, const float *hi_obs, const int32_t *lo, double *state, int64_t n, hipStream_t stream);
// true if launch_step(a) runs the fused rollout kernel, which also writes StepArgs::snap_* (other launches: the caller copies)
bool launch_step_is_rollout(int param_mode, const StepArgs &a);
// true if launch_step(a) with a.obs_part set runs a STATS instantiation (fused observation sums); envs per leaf of those partials
bool launch_step_supports_stats(int env_id, int param_mode, const StepArgs &a);
int64_t stats_leaf_envs(int env_id);
hipError_t launch_reset(int env_id, const ResetArgs &a, hipStream_t stream);
hipError_t launch_sample(int env_id, int param_mode, const SampleArgs &a, hipStream_t stream);
hipError_t launch_mixed_rollout(const MixedArgs &m, hipStream_t stream);
int64_t compact_chunks(int64_t n);
hipError_t launch_compact_final(int obs_dim, const CompactArgs &a, hipStream_t stream);
hipError_t launch_write_probe(float *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t n, int K, hipStream_t stream);
hipError_t launch_write_probe_env(int env_id, int flags, float *obs, void *rew, void *act, uint8_t *term, uint8_t *trunc, int64_t n, int K,
                                  hipStream_t stream);
hipError_t launch_set_word(uint64_t *dst, uint64_t value, hipStream_t stream);
hipError_t launch_add_word(uint64_t *dst, uint64_t delta, hipStream_t stream);   // *dst += delta (two's complement: a negative delta subtracts)

}  // namespace mxv
