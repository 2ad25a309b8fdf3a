/*
 * mxv.h — C ABI of the MI355X-native vectorised classic-control environment engine.
 *
 * What this boundary replaces.  openai/gym 0.26.2 has no FFI: its hot path
 * gym.vector.SyncVectorEnv.step()/reset() (gym/vector/sync_vector_env.py:90-169) is a serial
 * Python loop over TimeLimit-wrapped sub-envs (gym/wrappers/time_limit.py:39-68) whose step()
 * bodies are the classic-control dynamics (gym/envs/classic_control/{cartpole.py:130-188,
 * pendulum.py:119-139, acrobot.py:196-277, mountain_car.py:127-148,
 * continuous_mountain_car.py:142-175}).  The entry points below are what a ctypes binding
 * behind a gym.vector.VectorEnv subclass (gym/vector/vector_env.py:12-274) binds; the
 * reference-side stub is shown in INTEGRATION.md, the shipped one is gym_amd/_native.py.
 *
 * Conventions.  C linkage, plain pointers and sizes, no C++/torch types.  Every function
 * returns an int status (MXV_OK or a negative MXV_ERR_*); mxv_last_error() gives the message.
 * One handle = one HIP device + one HIP stream + the device-resident env state
 * (struct-of-arrays fp64 state[S][N], int32 elapsed[N]).  The library owns state buffers;
 * the CALLER owns every I/O buffer it passes.  Pointers suffixed _dev are device pointers
 * valid on the handle's device; _host are host pointers.  Device-pointer calls are
 * asynchronous on the handle's stream; *_host calls synchronise.  A handle is not
 * thread-safe; distinct handles are independent.
 *
 * Output dtypes follow SyncVectorEnv (sync_vector_env.py:65-72): observations float32
 * [N][O] row-major, rewards float64 [N] (float32 with MXV_FLAG_REWARD_F32), terminated /
 * truncated uint8 [N] (numpy bool layout).  Discrete actions are int64 [N] (the dtype of
 * batch_space(Discrete) = MultiDiscrete, gym/vector/utils/spaces.py:53-68) or int32 with
 * MXV_FLAG_ACTION_I32; Box actions are float32 [N] (= [N][1]).
 *
 * RNG contract (Philox4x32-10, counter based) — see DESIGN.md §RNG.  g = global_env >> 2, word = out[global_env & 3]:
 *   actions, Discrete(3) / Box : key = action_seed, ctr = (g_lo, g_hi, t_lo, (t_hi & 0x0fffffff) | 1<<28), one word per env and
 *             step t;  Discrete(n): (word*n)>>32 ;  Box(lo,hi): float32(lo + (hi-lo)*(word+0.5)*2^-32)
 *   actions, Discrete(2)       : a uniform action is one random bit: key = action_seed, ctr = (g_lo, g_hi, b_lo, (b_hi &
 *             0x0fffffff) | 6<<28) with b = t >> 5 (one call per block of 32 steps), action = (word >> (t & 31)) & 1
 *   resets  : key = per-env seed (base_seed + global_env unless explicit), ctr = (k, 0, 0, 2<<28), k = 0, 1, 2, ... = how many
 *             resets (explicit reset() or autoreset inside a vector step) this env has had since the last mxv_seed(): each env
 *             consumes its own reset stream in order, like the per-env generator of the reference (cartpole.py:202).  The
 *             ordinals are device state (uint32 [N]): mxv_get_episodes / mxv_set_episodes for checkpoints.
 *             uniform(low, high) = low + (high-low)*(word+0.5)*2^-32 in fp64, one word per state component.
 * t = index of the vector step since the last mxv_seed().  Streams use GLOBAL env indices: any sharding of one logical vector env
 * over several handles / GPUs draws the same numbers.
 *
 * Numerical contract (what "matches the reference" means at this boundary; tests/helpers.py holds the same bars).  On identical
 * fp64 states and actions, against gym 0.26.2 under NumPy 2.x + glibc:
 *   * terminated, truncated, elapsed steps, sampled discrete actions and Philox reset states are BIT-EXACT.  Acrobot's termination
 *     test `-cos(th1) - cos(th2 + th1) > 1.0` (acrobot.py:232-235) is decided by the last bit of sin / cos when the height is within
 *     an ulp of 1.0; there (whenever the engine's own height is within 2^-40 of 1.0: ~1 env-step in 10^12) the whole step is
 *     evaluated as the reference writes it on CORRECTLY ROUNDED sin / cos (gym_amd/csrc/mxv_exact.hpp), and the mask is the
 *     reference's on a correctly rounded libm — for all 4096 states of tests/golden/Acrobot_p1_threshold.npz, which straddle the
 *     threshold from 0 to 15 000 ulps.  The reference's glibc run differs from THAT in 2 of those states (heights that round to
 *     exactly 1.0, where glibc's cos is a neighbour of the rounded value, as it is for ~0.1 % of arguments): the one place where the
 *     reference's mask is a property of its libm build rather than of its arithmetic;
 *   * observations agree within 2 float32 ulps (inside north_star's fp32 rtol = 1e-5), not bit for bit: the fp64 state agrees to
 *     rtol 1e-12 and the last fp64 bit of sin/cos can move a float32 rounding;
 *   * rewards agree to rtol 1e-13 (Pendulum + 1e-9 absolute: the reference's u**2 is libm powf, not correctly rounded);
 *   * seeded streams do NOT match the reference's: it draws from PCG64, this engine from Philox4x32-10 (north_star); distributions do
 *     (tests/test_gpu_distributions.py).
 */
#ifndef MXV_H
#define MXV_H

/* API levels — additive: a consumer written against level n keeps working.  MXV_API_LEVEL is the level of this header.
 *   1  (round 1)  static information, lifetime, seeding, reset, step, rollout, host-buffer convenience, state access, physics parameters,
 *                 stream / sync: the ~25 entry points SURVEY.md §8(b) sketches — all a drop-in VectorEnv needs
 *   2  (round 2)  episode statistics, mapped / packed / block host I/O, action tapes, mixed-batch launch; mxv_norm.h, mxv_toytext.h
 *   3  (round 3)  collectives (mxv_comm.h), per-env parameters, final-tensor snapshots, mxv_wait_stream
 *   4  (round 4)  device clock (hipGraph capture), fused moments (mxv_set_obs_partials / mxv_set_return_partials); mxv_diag.h
 *   5  (round 5)  mxv_get_beyond / mxv_set_beyond, mxv_bj_rollout_compact; Blackjack's one-call draw contract — NOT additive: the card and
 *                 Discrete(2) action streams of a Blackjack handle changed, so its snapshots carry the contract number (mxv_toytext.h)
 *   6  (round 6)  mxv_subnorm_* (per-sub-env Normalize*, mxv_norm.h); mxv_adopt_obs (the observation buffer carries the state between single
 *                 steps); episode statistics of the toy_text engines (mxv_tab_* / mxv_bj_episode_stats ..., mxv_toytext.h); elapsed[] stored in
 *                 16 bits where the TimeLimit fits (no ABI change); caller tensors must be aligned to the width they are moved with
 *                 (MXV_ERR_INVALID_ARG otherwise — found by the fuzz harness, tests/test_abi_fuzz.py)
 * Every section below says the level it appeared at. */
#define MXV_API_LEVEL 6

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxv_handle mxv_handle;

/* env kinds (gym ids: CartPole-v0/v1, Pendulum-v1, Acrobot-v1, MountainCar-v0, MountainCarContinuous-v0) */
enum {
    MXV_CARTPOLE = 0,
    MXV_PENDULUM = 1,
    MXV_ACROBOT = 2,
    MXV_MOUNTAINCAR = 3,
    MXV_MOUNTAINCAR_CONT = 4,
    MXV_NUM_ENV_KINDS = 5
};

/* status codes */
enum {
    MXV_OK = 0,
    MXV_ERR_INVALID_ARG = -1,    /* bad config / NULL pointer / bad reset bounds (ValueError, classic_control/utils.py:41-44) */
    MXV_ERR_HIP = -2,            /* HIP runtime failure; message holds hipGetErrorString */
    MXV_ERR_INVALID_ACTION = -3, /* discrete action outside [0,n): the reference's `assert action_space.contains(action)` (cartpole.py:131-132) */
    MXV_ERR_RESET_NEEDED = -4,   /* step before reset: gym.error.ResetNeeded (gym/wrappers/order_enforcing.py:33-37) */
    MXV_ERR_UNSUPPORTED = -5
};

/* config flags */
enum {
    MXV_FLAG_ACTION_I32 = 1,   /* discrete actions are int32 instead of int64 */
    MXV_FLAG_REWARD_F32 = 2,   /* rewards are float32 instead of float64 */
    MXV_FLAG_NO_AUTORESET = 4  /* dynamics + TimeLimit only; finished envs are NOT reset (single-env semantics).  CartPole then keeps the
                                  reference's steps_beyond_terminated bookkeeping (cartpole.py:169-184): the step in which the pole falls
                                  pays 1.0, every later step of that env that is still terminated pays 0.0, until mxv_reset /
                                  mxv_set_state touches the env (the reference's one-time logger.warn is not reproduced) */
};

#define MXV_MAX_PARAMS 12
#define MXV_ENV_ALIGN 4 /* env_offset must be a multiple of this (Philox action groups of 4 envs) */
#define MXV_MAX_NUM_ENVS (1LL << 28) /* envs per handle (268 M); larger vector envs are shards: one handle per env_offset range */

/*
 * Physics parameter vector P[MXV_MAX_PARAMS] (fp64) = the attributes the reference's env objects hold
 * (VectorEnv.set_attr/get_attr, sync_vector_env.py:192-214, e.g. "gravity"):
 *  CartPole   : 0 gravity 1 masscart 2 masspole 3 total_mass 4 length 5 polemass_length 6 force_mag 7 tau
 *               8 theta_threshold_radians 9 x_threshold 10 kinematics_integrator (0 euler, 1 semi-implicit)
 *  Pendulum   : 0 max_speed 1 max_torque 2 dt 3 g 4 m 5 l
 *  Acrobot    : 0 dt 1 LINK_LENGTH_1 2 LINK_LENGTH_2 3 LINK_MASS_1 4 LINK_MASS_2 5 LINK_COM_POS_1 6 LINK_COM_POS_2
 *               7 LINK_MOI 8 MAX_VEL_1 9 MAX_VEL_2 10 torque_noise_max (> 0: torque += uniform(-m, m) from the
 *               step-noise stream: key = env seed, ctr = (t_lo, t_hi, 0, 4 << 28), word x) 11 book_or_nips (0 book, 1 nips)
 *  MountainCar: 0 min_position 1 max_position 2 max_speed 3 goal_position 4 goal_velocity 5 force 6 gravity
 *  MountainCarContinuous: 0 min_action 1 max_action 2 min_position 3 max_position 4 max_speed 5 goal_position
 *               6 goal_velocity 7 power
 */

typedef struct mxv_config {
    int32_t env_id;            /* MXV_CARTPOLE ... */
    int32_t device;            /* HIP device ordinal */
    int64_t num_envs;          /* envs held by THIS handle (a shard of the logical vector env) */
    int64_t env_offset;        /* global index of local env 0; RNG streams use global indices so that
                                  1/2/4/8-GPU shardings of one logical env draw identical numbers */
    int32_t max_episode_steps; /* TimeLimit (gym/envs/__init__.py:11-50); <= 0 disables truncation */
    int32_t flags;             /* MXV_FLAG_* */
    uint64_t seed;             /* base seed: env i is seeded with seed + global_index (sync_vector_env.py:106-107) */
    uint64_t action_seed;      /* seed of the action-sampling stream (VectorEnv.action_space.seed) */
} mxv_config;

/* -- static information ---------------------------------------------------------------------- */
/* state dim S, observation dim O, number of discrete actions (0 for Box action spaces). */
int mxv_env_dims(int32_t env_id, int32_t *state_dim, int32_t *obs_dim, int32_t *num_actions);
/* default P[] (the values the reference's __init__ sets) and default reset bounds (low, high)
 * (Pendulum: (x_init, y_init) = (pi, 1), pendulum.py:14-15,141-159). */
int mxv_default_params(int32_t env_id, double *params_host);
int mxv_default_reset_bounds(int32_t env_id, double *bounds2_host);
const char *mxv_version(void);

/* -- lifetime -------------------------------------------------------------------------------- */
int mxv_create(const mxv_config *cfg, mxv_handle **out);
int mxv_destroy(mxv_handle *h);
/* h may be NULL: message of the last failed mxv_create on this thread. */
const char *mxv_last_error(const mxv_handle *h);

/* -- seeding (Env.reset(seed=...), gym/core.py:149-151; SyncVectorEnv seeds env i with seed+i) -- */
/* per_env_seeds_host: NULL -> env i uses base_seed + env_offset + i; else N explicit 64-bit seeds.
 * Restarts the step index t and every env's reset ordinal at 0. */
int mxv_seed(mxv_handle *h, uint64_t base_seed, const uint64_t *per_env_seeds_host);
int mxv_seed_actions(mxv_handle *h, uint64_t action_seed);

/* -- reset: SyncVectorEnv.reset_wait (sync_vector_env.py:90-129) + TimeLimit.reset ------------ */
/* mask_dev: NULL = all envs, else uint8[N] (1 = reset this env).  bounds2_host: NULL = defaults,
 * else (low, high) from reset(options={"low","high"}) (classic_control/utils.py:17-46; Pendulum:
 * (x_init, y_init)); low > high -> MXV_ERR_INVALID_ARG.  obs_dev may be NULL. */
int mxv_reset(mxv_handle *h, const uint8_t *mask_dev, const double *bounds2_host, float *obs_dev);

/* -- step: SyncVectorEnv.step_wait (sync_vector_env.py:135-169), TimeLimit + autoreset fused ---- */
/* On terminated|truncated (and unless MXV_FLAG_NO_AUTORESET): obs row = post-reset observation,
 * final_obs row = terminal observation (info["final_observation"]); rows of envs that did not
 * finish are left untouched in final_obs.  final_obs_dev may be NULL.  The mask of finished envs
 * is terminated|truncated.  An out-of-range discrete action leaves that env unstepped and latches
 * MXV_ERR_INVALID_ACTION, reported by the next mxv_sync()/ *_host call. */
int mxv_step(mxv_handle *h, const void *actions_dev, float *obs_dev, void *reward_dev,
             uint8_t *terminated_dev, uint8_t *truncated_dev, float *final_obs_dev);
/* Same, with actions drawn on device from the Philox action stream (action_space.sample(),
 * gym/spaces/{discrete.py:81,multi_discrete.py:123,box.py:216-222}); actions_out_dev (may be
 * NULL) receives the actions taken, in the action dtype. */
int mxv_step_sampled(mxv_handle *h, void *actions_out_dev, float *obs_dev, void *reward_dev,
                     uint8_t *terminated_dev, uint8_t *truncated_dev, float *final_obs_dev);
/* K sampled steps back to back (the user loop of README.md:29-41 with a random policy).
 * per_step != 0: every output pointer addresses [K][...] and step k writes slice k (trajectory
 * buffers); per_step == 0: every output pointer addresses one step's buffers, overwritten K times
 * (the "final tensors" of a rollout chunk).  Any output pointer may be NULL except obs_dev.
 * mode: MXV_ROLLOUT_EAGER = K launches of the step kernel; MXV_ROLLOUT_GRAPH = the same K launches
 * replayed from a cached hipGraph; MXV_ROLLOUT_FUSED = ONE launch that keeps every env's state in
 * registers across the K steps (state/elapsed touch HBM once per chunk instead of once per step).
 * All three produce bit-identical results. */
enum { MXV_ROLLOUT_EAGER = 0, MXV_ROLLOUT_GRAPH = 1, MXV_ROLLOUT_FUSED = 2 };
int mxv_rollout(mxv_handle *h, int32_t K, int32_t per_step, int32_t mode, void *actions_out_dev,
                float *obs_dev, void *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev,
                float *final_obs_dev);
/* K steps in one fused launch driven by an action tape actions_tape_dev[K][N] (action dtype of the
 * handle) instead of the Philox action stream: scripted / policy-chunk rollouts. */
int mxv_rollout_tape(mxv_handle *h, int32_t K, int32_t per_step, const void *actions_tape_dev,
                     float *obs_dev, void *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev,
                     float *final_obs_dev);
/* Second destination for the FINAL tensors of every following mxv_rollout / mxv_rollout_tape / mxv_rollout_mixed call: the
 * last of the K steps also writes its obs / reward / terminated / truncated (the handle's dtypes, [N] each; reward .. truncated
 * may be NULL) into these device buffers — by the fused kernel itself, or by device-to-device copies on the handle's stream
 * for the other launch modes.  This is the snapshot a sharded vector env hands to mxv_allgather_outputs while the next chunk
 * runs: no copy kernels between rollout and gather, and the trajectory tensors may be reused at once.  NULL obs detaches. */
int mxv_set_final_snapshot(mxv_handle *h, float *obs_dev, void *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev);
/* Heterogeneous dispatch (BASELINE.json configs[4]: mixed batch {CartPole, Pendulum, Acrobot, MountainCar}).  The reference has
 * no mixed vector env (gym/vector/vector_env.py:20-23; SyncVectorEnv requires identical sub-env spaces, sync_vector_env.py:
 * 220-234): a mixed batch IS a set of homogeneous vector envs, here one handle per segment on one device.  This call advances
 * all of them by K sampled steps in ONE kernel launch (a block -> segment table; every wave runs the rollout body of its
 * segment's env kind), bit-identical to calling mxv_rollout(FUSED) on each handle.  outs[i] = the output pointers of handles[i]
 * (meaning of mxv_rollout's; any but obs may be NULL).  The launch goes to handles[0]'s stream; the other handles' streams are
 * ordered around it on the GPU.  MXV_ERR_UNSUPPORTED if a segment runs non-default physics attributes / without autoreset. */
#define MXV_MAX_MIXED 8
typedef struct mxv_step_outputs {
    void *actions_out;
    float *obs;
    void *reward;
    uint8_t *terminated;
    uint8_t *truncated;
    float *final_obs;
} mxv_step_outputs;
int mxv_rollout_mixed(mxv_handle *const *handles, int32_t count, int32_t K, int32_t per_step, const mxv_step_outputs *outs);
/* action_space.sample() for the NEXT step index without stepping. */
int mxv_sample_actions(mxv_handle *h, void *actions_out_dev);
/* -- host-buffer convenience (what a NumPy-returning gym.vector.VectorEnv adapter calls) -------- */
/* Copies through library-owned staging buffers and synchronises.  final_obs_host may be NULL. */
int mxv_reset_host(mxv_handle *h, const uint8_t *mask_host, const double *bounds2_host, float *obs_host);
int mxv_step_host(mxv_handle *h, const void *actions_host, float *obs_host, void *reward_host,
                  uint8_t *terminated_host, uint8_t *truncated_host, float *final_obs_host);

/* Zero-copy form of the two calls above (SyncVectorEnv(copy=False), sync_vector_env.py:61-63,163: "return the internal
 * buffer"): mxv_host_io allocates ONE block of pinned, device-mapped host memory holding the step I/O of all envs and returns
 * host pointers into it (actions in the handle's action dtype, obs/final_obs float32 [N][O], reward in the handle's reward
 * dtype, flags uint8 [N]); mxv_step_mapped / mxv_reset_mapped take the actions from and leave the outputs in that block,
 * then synchronise.  Small envs (block <= 2 MiB): the kernel itself reads/writes the block over PCIe — one launch, no
 * copies; larger envs: device staging plus one DMA copy each way (1-byte flag stores over PCIe would throttle the kernel).
 * The buffers are overwritten by the next call.  Any out-pointer may be NULL. */
int mxv_host_io(mxv_handle *h, void **actions, float **obs, void **reward, uint8_t **terminated, uint8_t **truncated,
                float **final_obs);
int mxv_step_mapped(mxv_handle *h);
int mxv_reset_mapped(mxv_handle *h, const double *bounds2_host);

/* info["final_observation"] for host callers of LARGE vector envs (step I/O above 2 MiB).  Only the rows of the envs that finished a
 * step mean anything (sync_vector_env.py:152-156), typically a few percent of the batch, so the host calls never move the dense
 * [N][O] array over PCIe: the device packs (env index, row) pairs in ascending env order (two small kernels, no atomics), two small DMAs
 * bring them over, and by default the library scatters them into the caller's dense final_obs array — 0.5 ms of cache misses per step
 * at 2^20 envs.  mxv_final_packed(h, 1, &supported) skips the scatter: after every mxv_step_host / mxv_step_mapped the pairs
 * of THAT step are read through mxv_final_packed_view (pointers into the library's pinned buffer, valid until the next step:
 * *count pairs, idx[i] = env index, rows[i*O .. i*O+O) = its terminal observation; idx ascending = np.flatnonzero(terminated | truncated)); final_obs_host is then
 * ignored and the mapped block's final_obs region is not updated.  supported = 0 for small envs (nothing to pack: the kernel
 * writes the pinned block itself); the dense path stays in force there. */
int mxv_final_packed(mxv_handle *h, int32_t enable, int32_t *supported);
int mxv_final_packed_view(mxv_handle *h, const int32_t **count, const int32_t **idx, const float **rows);
/* With episode statistics enabled (mxv_episode_stats), the packed record of a host step also carries what
 * RecordEpisodeStatistics reports for the finished envs (record_episode_statistics.py:124-143): ep_return[i] / ep_length[i] =
 * return (float32 accumulator) and length of the episode that ended at env idx[i] — instead of two dense [N] arrays
 * (mxv_episode_stats_host) of which a few percent mean anything.  Same lifetime as the other views. */
int mxv_final_packed_stats_view(mxv_handle *h, const float **ep_return, const int32_t **ep_length);

/* One-DMA form of mxv_step_host: the step's outputs land in ONE caller-supplied host block
 *     final_obs float32 [N][O] | obs float32 [N][O] | reward | terminated uint8 [N] | truncated uint8 [N]
 * (each region padded to 256 B; mxv_host_block_layout returns the block size and the region offsets) with a single
 * device-to-host copy of obs .. truncated instead of four — 26 MB in one 0.46-ms DMA at 2^20 CartPole envs, where four
 * copies into separate pageable arrays cost 0.65 ms.  mxv_host_alloc returns pinned memory for such blocks (DMA without
 * staging; any pointer works, pinned is faster); the NumPy adapter keeps a small pool of them and hands out views, a block
 * being reused only once the caller dropped every array of it.  want_final = 0: final_obs is not produced.  want_final != 0:
 * with mxv_final_packed enabled the rows of the finished envs are left packed (mxv_final_packed_view) and the block's
 * final_obs region is not written; otherwise the dense rows are part of the same DMA.  Synchronises; errors as mxv_step_host. */
/* Where the outputs of the LAST host step (or host reset: obs only) still sit in memory the GPU can read — the device staging
 * of a large env, the pinned block of a small one — as addresses valid on the handle's device until the next host call.  What a
 * device-side consumer of a host loop starts from (NormalizeObservation / NormalizeReward stacked on the NumPy adapter
 * normalise these instead of uploading the arrays the step just downloaded).  obs float32 [N][O], reward in the handle's
 * reward dtype, flags uint8 [N]; any out-pointer may be NULL. */
int mxv_staging_view(mxv_handle *h, const float **obs, const void **reward, const uint8_t **terminated, const uint8_t **truncated);
int mxv_host_alloc(size_t bytes, void **ptr);
int mxv_host_free(void *ptr);
int mxv_host_block_layout(mxv_handle *h, size_t *bytes, size_t *final_obs_off, size_t *obs_off, size_t *reward_off,
                          size_t *terminated_off, size_t *truncated_off);
int mxv_step_host_block(mxv_handle *h, const void *actions_host, void *block_host, int32_t want_final);

/* -- state access (parity hook + checkpoint/resume) ---------------------------------------------- */
/* -- "the observation carries the state" (API level 6): mxv_step moves the fp64 state both ways every launch (16 S of its ~108 bytes per
 * env-step) because fp32 state fails the parity bar.  For CartPole and both MountainCars the observation IS float32(state), and the
 * remainder state - float32(state) is an exact int32 multiple of 2^(exponent - 53): a handle that ADOPTS the caller's observation buffer
 * keeps the state as (that buffer, int32 residuals) between single steps — 12 instead of 16 bytes per state component per step, same
 * bits (values outside float32's normal range escape to the fp64 array).  The contract is the caller's: between two mxv_step /
 * mxv_step_sampled calls that pass obs_dev == the adopted buffer, the buffer must not be written (it is half of the state), and it must
 * outlive the adoption; every other call (reset, rollouts, get / set_state, steps with another obs pointer or non-default parameters)
 * first brings the state back into the fp64 array by itself.  obs_dev = NULL releases.  MXV_ERR_UNSUPPORTED for Pendulum / Acrobot. */
int mxv_adopt_obs(mxv_handle *h, float *obs_dev);
/* state_soa_host: double[S][N]; elapsed_host: int32[N]; either may be NULL.  Synchronises. */
int mxv_get_state(mxv_handle *h, double *state_soa_host, int32_t *elapsed_host);
int mxv_set_state(mxv_handle *h, const double *state_soa_host, const int32_t *elapsed_host);
/* step index t (position of the action stream) / number of explicit reset calls since seeding (informational) — restore with
 * mxv_set_counters when resuming. */
int mxv_get_counters(mxv_handle *h, uint64_t *t, uint32_t *r);
int mxv_set_counters(mxv_handle *h, uint64_t t, uint32_t r);
/* Device clock: the vector-step index (counter of the action and step-noise streams) normally travels to every launch as a kernel
 * argument, which a hipGraph would freeze at its capture-time value.  With the device clock on, every launch of this handle reads the
 * index from a device word and a one-thread kernel behind it advances the word — both are ordinary stream work, so a caller may
 * RECORD calls of this handle into its own hipGraph (hipStreamBeginCapture / torch.cuda.graph on the handle's stream, or on a stream
 * the handle's stream is joined to) — mxv_step with a policy in between, mxv_step_sampled, mxv_rollout, mxv_rollout_tape — and replay
 * the graph any number of times: the replays continue the streams exactly where single calls would (tests/test_gpu_graph_capture.py:
 * replayed graphs == the same calls made one by one, bit for bit, including Acrobot's step-indexed torque noise).  The host's copy of
 * the index is refreshed from the device by mxv_get_counters (which then synchronises the stream).  on = 0 reads the index back and
 * returns to argument passing.  Calls that synchronise or copy to the host (the *_host calls, mxv_sync, mxv_get_state) cannot be
 * captured, as with any stream.  WITHOUT the device clock a step / rollout call on a stream that is being captured returns
 * MXV_ERR_UNSUPPORTED (its replays would repeat one step index, i.e. the same action and noise draws).  What a recording freezes
 * besides the index: seeds, reset bounds, parameters and the kernel variant (guarded / unguarded) — re-record after mxv_seed*,
 * reset(options) bounds, mxv_set_params* or an mxv_set_state outside the unguarded range. */
int mxv_set_device_clock(mxv_handle *h, int32_t on);
/* NormalizeObservation's batch moments, fused into the rollout (gym/wrappers/normalize.py:17-29: every step's batch mean / var are the
 * only cross-env reduction on the path).  With a buffer attached, every sampled trajectory launch (mxv_rollout, MXV_ROLLOUT_FUSED,
 * per_step != 0, K >= 2, all per-step outputs, default physics parameters) also leaves, for each of its K steps and each tile of
 * envs_per_leaf consecutive envs, the fp64 column sums and sums of squares of the observations it wrote:
 * partials_dev[K][leaves][values], values = 2 O (sums, then sums of squares) — 0.5 B per env-step instead of a second pass that reads
 * the 4 O bytes back.  mxv_norm_obs_sums_partials folds them (fixed binary tree: bit-reproducible, the same for any power-of-two
 * sharding) into the [K][2 O] sums mxv_norm_obs_apply takes.  A launch that cannot produce them fails with MXV_ERR_UNSUPPORTED
 * (nothing is skipped silently); NULL detaches.  The caller owns the buffer. */
int mxv_set_obs_partials(mxv_handle *h, double *partials_dev);
int mxv_obs_partials_layout(mxv_handle *h, int64_t *leaves, int64_t *envs_per_leaf, int32_t *values);
/* The same for NormalizeReward (normalize.py:127-145): the running discounted returns `returns = returns * gamma + rews`, zeroed where
 * an episode ended, are advanced by the rollout in registers (returns_state_dev [N] float64: read at entry, written at exit — the array
 * mxv_norm_returns_ptr gives), and every step's per-tile sum and sum of squares of the updated returns are left in
 * partials_dev[K][leaves][2] for mxv_norm_reward_sums_partials.  Same launches, same leaves as above; both may be attached. */
int mxv_set_return_partials(mxv_handle *h, double *returns_state_dev, double gamma, double *partials_dev);
/* per-env reset ordinals (position of each env's reset stream, see RNG contract): uint32[N].  Synchronises. */
int mxv_get_episodes(mxv_handle *h, uint32_t *episodes_host);
int mxv_set_episodes(mxv_handle *h, const uint32_t *episodes_host);
/* CartPole's steps_beyond_terminated marks (cartpole.py:169-184; kept only with MXV_FLAG_NO_AUTORESET): uint8[N], 1 = this env has
 * terminated since its last reset and pays 0.0 from now on.  Part of a checkpoint: mxv_set_state clears the marks (a fresh state), so
 * restore them AFTER it.  Handles without marks: get fills zeros, set accepts zeros only. */
int mxv_get_beyond(mxv_handle *h, uint8_t *beyond_host);
int mxv_set_beyond(mxv_handle *h, const uint8_t *beyond_host);

/* -- physics parameters (VectorEnv.get_attr/set_attr/call, sync_vector_env.py:171-214) ----------- */
/* One value per attribute for all sub-envs (set_attr with a scalar or a list of equal values).  Default values run
 * the kernels with the constants folded in; any other value switches the handle to the runtime-parameter kernels. */
int mxv_get_params(mxv_handle *h, double *params_host);
int mxv_set_params(mxv_handle *h, const double *params_host);
/* A value per sub-env (set_attr with a list of differing values, e.g. env.set_attr("gravity", [9.81, 3.72, 8.87, 1.62]),
 * tests/vector/test_sync_vector_env.py:101-110): params_host is double[MXV_MAX_PARAMS][N] (attribute-major).  The
 * handle then steps with the per-env-parameter kernels (one launch per step; the fused fast path needs equal
 * attributes) until mxv_set_params() sets common values again.  mxv_get_params_per_env always fills [MXV_MAX_PARAMS][N]. */
int mxv_set_params_per_env(mxv_handle *h, const double *params_host);
int mxv_get_params_per_env(mxv_handle *h, double *params_host);

/* -- episode statistics: gym.wrappers.RecordEpisodeStatistics (gym/wrappers/record_episode_statistics.py:96-151) fused
 *    into the step kernels: no extra pass over the step outputs ------------------------------------ */
/* enable != 0: allocate the per-env running-return accumulators (float32, like the reference's np.float32 array;
 * episode length is the TimeLimit counter) and zero them; they are zeroed again for every env an explicit reset
 * touches (:91-94).  Needs autoreset.  enable == 0 frees them. */
int mxv_episode_stats(mxv_handle *h, int32_t enable);
/* Attach device output buffers used by every following mxv_step* / mxv_rollout* call: float32 returns and int32
 * lengths, [N] (or [K][N] when the call is made with per_step != 0).  Entries are WRITTEN ONLY where
 * terminated | truncated of that step is set (that flag pair is the "_episode" mask, :33-35); other entries keep
 * their previous content.  Either pointer may be NULL. */
int mxv_set_episode_outputs(mxv_handle *h, float *ep_return_dev, int32_t *ep_length_dev);
/* Host view after mxv_step_host: returns / lengths of the episodes that ended in that step (valid where its
 * terminated | truncated is set) and the running returns of all envs (episode_returns).  Any pointer may be NULL. */
int mxv_episode_stats_host(mxv_handle *h, float *ep_return_host, int32_t *ep_length_host, float *running_return_host);
/* Checkpoint restore of the running returns read with mxv_episode_stats_host(running_return_host): float32[N].  Together
 * with mxv_set_state / mxv_set_counters / mxv_seed* / mxv_set_params* this rebuilds a handle that continues bit-identically
 * (the reference's envs are restored by pickling: tests/envs/test_envs.py:192-200). */
int mxv_set_running_returns(mxv_handle *h, const float *running_return_host);

/* -- stream / sync -------------------------------------------------------------------------------- */
/* Waits for the handle's stream; returns MXV_ERR_INVALID_ACTION if a step since the last
 * sync saw an out-of-range action (and clears the latch). */
int mxv_sync(mxv_handle *h);
/* The hipStream_t the handle launches on (created non-blocking by mxv_create) / adopt an external one. */
int mxv_get_stream(mxv_handle *h, void **stream);
int mxv_set_stream(mxv_handle *h, void *stream);
/* GPU-side ordering, no host wait: everything queued on `other_stream` (a hipStream_t; NULL = the default stream) so far
 * completes before anything the handle launches from now on — what a learner whose policy produced the actions on its own
 * stream calls before mxv_step (one event record + one stream wait; a no-op if it IS the handle's stream). */
int mxv_wait_stream(mxv_handle *h, void *other_stream);

#ifdef __cplusplus
}
#endif

/* The rest of the ABI lives in per-subsystem headers, included here so that `#include "mxv.h"` declares the whole product surface:
 *   mxv_norm.h     NormalizeObservation / NormalizeReward kernels (mxv_norm_*)
 *   mxv_toytext.h  tabular toy_text engine (mxv_tab_*) and Blackjack (mxv_bj_*)
 *   mxv_comm.h     RCCL collectives of a sharded vector env (mxv_comm_*, mxv_allgather_*)
 * and, NOT included here (optional; include it yourself):
 *   mxv_diag.h     diagnostics: mxv_last_launch, write / HBM-class probes, placed memory */
#include "mxv_norm.h"
#include "mxv_toytext.h"
#include "mxv_comm.h"

#endif /* MXV_H */
