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
/* Which kernel instantiation the handle's LAST step / rollout launch took — the library picks among ~150 (one launch per step or K
 * fused steps, envs per lane by shard size, guarded or unguarded trigonometry, folded or runtime physics parameters, the output-dtype
 * specialisations of the trajectory launch).  Lets a test or a benchmark state which code it measured (tests/test_gpu_soak.py,
 * bench.py config.launch_info).  kernel = -1 before the first launch; mxv_rollout_mixed does not update it. */
typedef struct mxv_launch_info {
    int32_t kernel;        /* 0 = step_kernel (one launch per step; also EAGER / GRAPH rollouts), 1 = rollout_kernel_v3 (FUSED) */
    int32_t env_id;
    int32_t param_mode;    /* 1 = default attributes folded into the code, 0 = common runtime values, 2 = per-env values */
    int32_t envs_per_lane;
    int32_t safe;          /* 1 = guarded sin / cos (state injected, unusual reset bounds, non-default attributes) */
    int32_t out_mode;      /* rollout_kernel_v3: 1 = trajectory outputs float64 rewards + int64 actions, 2 = float32 + int32, 0 = generic */
    int32_t tape;          /* actions supplied by the caller */
    int32_t steps;         /* K of the launch */
    uint32_t grid, block;
} mxv_launch_info;
int mxv_last_launch(mxv_handle *h, mxv_launch_info *out);

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

/* -- running normalisation: gym.wrappers.NormalizeObservation / NormalizeReward (gym/wrappers/normalize.py:8-144), SURVEY.md
 *    §8f-2.  One mxv_norm = one RunningMeanStd (normalize.py:8-29: fp64 mean[dim], var[dim], count = 1e-4 at creation)
 *    plus, for rewards, the wrapper's per-env discounted-return accumulator (:123), device resident.  It works on the
 *    tensors the step calls produce, K consecutive batches per call ([K][num_envs][dim]; K = 1 for a single step()):
 *    for k: rms.update(batch_k) then the affine map with the UPDATED statistics, exactly the wrappers' order.  Batch
 *    moments are formed in the reference's dtypes (float32 for observations, float64 for returns) from exact-order fp64
 *    sums (fixed binary trees: bit-reproducible, and invariant under power-of-two sharding of the env axis).
 *    dim must be one of 1, 2, 3, 4, 6.  Calls are asynchronous on `stream` (a hipStream_t; NULL = null stream). -------- */
typedef struct mxv_norm mxv_norm;
int mxv_norm_create(int32_t device, int32_t dim, int64_t num_envs, void *stream, mxv_norm **out);
int mxv_norm_destroy(mxv_norm *nm);
const char *mxv_norm_last_error(const mxv_norm *nm); /* nm may be NULL: last failed mxv_norm_create on this thread */
int mxv_norm_set_stream(mxv_norm *nm, void *stream);
/* obs_rms.mean / .var / .count (and NormalizeReward.returns, double[num_envs]); any pointer may be NULL.  Synchronises. */
int mxv_norm_get_state(mxv_norm *nm, double *mean_host, double *var_host, double *count_host, double *returns_host);
int mxv_norm_set_state(mxv_norm *nm, const double *mean_host, const double *var_host, double count, const double *returns_host);
/* NormalizeObservation.normalize (:90-93): y[k] = (x[k] - mean) / sqrt(var + epsilon) after rms.update(x[k]).
 * x_dev float32 [K][num_envs][dim]; y_dev float64 (the reference's result dtype: float32 - float64) or float32 when
 * out_f32 != 0; y_dev may alias x_dev only when out_f32 != 0. */
int mxv_norm_observations(mxv_norm *nm, int32_t K, const float *x_dev, void *y_dev, int32_t out_f32, double epsilon);
/* NormalizeReward.step (:127-145): returns = returns*gamma + rews; return_rms.update(returns); out = rews / sqrt(var +
 * epsilon); returns[terminated | truncated] = 0.  reward/out are float64 [K][num_envs] (float32 when reward_f32 != 0);
 * out_dev may alias reward_dev.  Needs dim == 1. */
int mxv_norm_rewards(mxv_norm *nm, int32_t K, const void *reward_dev, int32_t reward_f32, const uint8_t *terminated_dev,
                     const uint8_t *truncated_dev, void *out_dev, double gamma, double epsilon);
/* Split form for a vector env sharded over `world` handles / GPUs (the batch of the reference is ALL num_envs rows):
 * *_sums writes this shard's per-step (sum_0..sum_{dim-1}, sumsq_0..sumsq_{dim-1}) to sums_dev[K][2*dim] (and, for
 * rewards, advances this shard's return accumulators); the caller concatenates the shards' sums in rank order
 * (all-gather) into all_sums_dev[world][K][2*dim]; *_apply merges them (binary tree over the rank index), runs the
 * running update with batch_count = total_rows and applies the map to this shard's rows.  world <= 64.
 * The one-call forms above are *_sums + *_apply with world = 1. */
int mxv_norm_obs_sums(mxv_norm *nm, int32_t K, const float *x_dev, double *sums_dev);
/* the same sums from partials a rollout left behind (mxv_set_obs_partials): [K][leaves][2 dim] -> sums_dev [K][2 dim] */
int mxv_norm_obs_sums_partials(mxv_norm *nm, int32_t K, const double *partials_dev, int64_t leaves, double *sums_dev);
int mxv_norm_reward_sums_partials(mxv_norm *nm, int32_t K, const double *partials_dev, int64_t leaves, double *sums_dev);
/* device address of the running discounted returns [N] (float64) this object keeps for NormalizeReward */
int mxv_norm_returns_ptr(mxv_norm *nm, double **returns_dev);
int mxv_norm_obs_apply(mxv_norm *nm, int32_t K, const float *x_dev, void *y_dev, int32_t out_f32, double epsilon,
                       const double *all_sums_dev, int32_t world, int64_t total_rows);
int mxv_norm_reward_sums(mxv_norm *nm, int32_t K, const void *reward_dev, int32_t reward_f32, const uint8_t *terminated_dev,
                         const uint8_t *truncated_dev, double gamma, double *sums_dev);
int mxv_norm_reward_apply(mxv_norm *nm, int32_t K, const void *reward_dev, int32_t reward_f32, void *out_dev, double epsilon,
                          const double *all_sums_dev, int32_t world, int64_t total_rows);

/* -- tabular toy_text environments: FrozenLake-v1 / FrozenLake8x8-v1 / Taxi-v3 / CliffWalking-v0 ---------
 *    One table-driven engine for the reference classes whose step() is `i = categorical_sample(P[s][a] probabilities);
 *    p, s, r, t = P[s][a][i]` and whose reset() is `categorical_sample(initial_state_distrib)` (gym/envs/toy_text/
 *    frozen_lake.py:247-270, taxi.py:254-278, cliffwalking.py:148-166, utils.py:4-8), with TimeLimit and SyncVectorEnv's
 *    autoreset fused as above.  The caller supplies the MDP as dense host tables [S][A][M] (M = longest transition list):
 *    cum_prob = np.cumsum of the list's probabilities, padded with -1; prob / next_state / reward / terminated per
 *    transition (padding ignored); initial_cum[S] = np.cumsum(initial_state_distrib).  Observations and actions are
 *    int64 [N] (MultiDiscrete, gym/vector/utils/spaces.py:53-68), rewards float64, info["prob"] float64.
 *    RNG: actions from the word-per-step Philox action stream above (ctr stream id 1, Discrete(A): (word*A)>>32); transitions from a Philox4x32-10 call keyed by the env's seed, ctr =
 *    (b_lo, b_hi, 0, 3 << 28), b = t >> 1: words (x, y) serve step 2b, (z, w) step 2b+1 — first the transition's uniform, then
 *    the uniform of an autoreset inside that step; explicit resets: key = env seed, ctr = (t_lo, t_hi, r, 2<<28), r = ordinal of the reset call (word x);
 *    uniform = (word + 0.5) * 2^-32. -------------------------------- */
typedef struct mxv_tab mxv_tab;
typedef struct mxv_tab_config {
    int32_t device;
    int32_t num_states;        /* S */
    int32_t num_actions;       /* A */
    int32_t max_transitions;   /* M */
    int64_t num_envs;
    int64_t env_offset;        /* global index of local env 0 (multiple of MXV_ENV_ALIGN) */
    int32_t max_episode_steps; /* TimeLimit; <= 0 disables (CliffWalking-v0 has none) */
    int32_t flags;             /* MXV_TAB_FLAG_* */
    uint64_t seed;
    uint64_t action_seed;
} mxv_tab_config;
/* MXV_TAB_FLAG_COMPACT: the trajectory calls (mxv_tab_rollout, mxv_tab_rollout_tape) take and produce the contract dtypes of SURVEY.md
 * §8(d) — int32 observations / actions (tape included), float32 rewards / probs: 18 B per env-step instead of 34 — on the device tensors;
 * every other call (mxv_tab_step, reset, the host calls) keeps the reference's int64 / float64.  Same values, narrower stores. */
enum { MXV_TAB_FLAG_COMPACT = 1, MXV_TAB_FLAG_GENERAL_KERNEL = 2 };
/* mxv_tab_rollout launches with every per-step output present (actions, obs, reward, both flags, prob; no final_* tensors) run a
 * kernel specialised for them (gym_amd/csrc/mxv_tab.hip: tab_traj_kernel — categorical_sample as integer compares against thresholds
 * packed at create time) whenever the MDP allows the packing: transition lists of length 1 or 3 whose cumulative probabilities end at
 * 1, float32-representable rewards, 64 KiB of table at most.  Same streams, same values, bit for bit.  MXV_TAB_FLAG_GENERAL_KERNEL
 * keeps such a handle on the general kernel (the tests' A/B switch); mxv_tab_last_kernel reports which one the last step / rollout
 * call launched. */
enum { MXV_TAB_KERNEL_NONE = 0, MXV_TAB_KERNEL_GENERAL = 1, MXV_TAB_KERNEL_TRAJECTORY = 2 };
int mxv_tab_create(const mxv_tab_config *cfg, const double *cum_prob_host, const double *prob_host,
                   const int32_t *next_state_host, const double *reward_host, const uint8_t *terminated_host,
                   const double *initial_cum_host, mxv_tab **out);
int mxv_tab_destroy(mxv_tab *h);
const char *mxv_tab_last_error(const mxv_tab *h);
int mxv_tab_seed(mxv_tab *h, uint64_t base_seed, const uint64_t *per_env_seeds_host);
int mxv_tab_seed_actions(mxv_tab *h, uint64_t action_seed);
/* mask_dev NULL = all envs; obs_dev (may be NULL) receives the states as int64. */
int mxv_tab_reset(mxv_tab *h, const uint8_t *mask_dev, int64_t *obs_dev);
/* One vector step.  uniforms_dev: NULL = Philox; else double[2][N] = (transition uniform, autoreset uniform) per env — the
 * values the reference's np_random.random() returned, for bit-exact replays.  On terminated | truncated: obs = the reset
 * state, prob = 1.0 (reset()'s info), final_obs / final_prob = the terminal state and its transition probability
 * (info["final_observation"], info["final_info"]["prob"]; rows of other envs untouched).  Any output but obs_dev may be NULL. */
int mxv_tab_step(mxv_tab *h, const int64_t *actions_dev, const double *uniforms_dev, int64_t *obs_dev, double *reward_dev,
                 uint8_t *terminated_dev, uint8_t *truncated_dev, double *prob_dev, int64_t *final_obs_dev,
                 double *final_prob_dev);
/* K steps in ONE launch (state + TimeLimit counter in registers), actions sampled on device (Discrete(A).sample()) or read
 * from a tape int64 [K][N]; per_step != 0: outputs are [K][N] trajectories, else overwritten K times.  Integer tensors are int64 and
 * real ones float64 — int32 / float32 with MXV_TAB_FLAG_COMPACT. */
int mxv_tab_rollout(mxv_tab *h, int32_t K, int32_t per_step, void *actions_out_dev, void *obs_dev, void *reward_dev,
                    uint8_t *terminated_dev, uint8_t *truncated_dev, void *prob_dev, void *final_obs_dev,
                    void *final_prob_dev);
int mxv_tab_rollout_tape(mxv_tab *h, int32_t K, int32_t per_step, const void *actions_tape_dev, void *obs_dev,
                         void *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, void *prob_dev,
                         void *final_obs_dev, void *final_prob_dev);
/* host-buffer convenience (staged copies, synchronising) */
int mxv_tab_reset_host(mxv_tab *h, const uint8_t *mask_host, int64_t *obs_host);
int mxv_tab_step_host(mxv_tab *h, const int64_t *actions_host, const double *uniforms_host, int64_t *obs_host,
                      double *reward_host, uint8_t *terminated_host, uint8_t *truncated_host, double *prob_host,
                      int64_t *final_obs_host, double *final_prob_host);
/* env.unwrapped.s and TimeLimit._elapsed_steps: int32 [N] each (either may be NULL) */
int mxv_tab_get_state(mxv_tab *h, int32_t *state_host, int32_t *elapsed_host);
int mxv_tab_set_state(mxv_tab *h, const int32_t *state_host, const int32_t *elapsed_host);
int mxv_tab_get_counters(mxv_tab *h, uint64_t *t, uint32_t *r);
int mxv_tab_set_counters(mxv_tab *h, uint64_t t, uint32_t r);
int mxv_tab_sync(mxv_tab *h);
/* the device clock of mxv_set_device_clock for this engine: mxv_tab_step / mxv_tab_rollout / mxv_tab_rollout_tape become recordable in a
 * caller's hipGraph (explicit resets are not: their ordinal travels by value) */
int mxv_tab_set_device_clock(mxv_tab *h, int32_t on);
int mxv_tab_last_kernel(const mxv_tab *h);
/* The integer form of categorical_sample's comparison (host function, no device needed): T in [0, 2^32] with
 * cum_prob > (w + 0.5) * 2^-32  <=>  w < T  for every 32-bit word w. */
uint64_t mxv_tab_word_threshold(double cum_prob);
int mxv_tab_set_stream(mxv_tab *h, void *stream);

/* -- Blackjack-v1 (gym/envs/toy_text/blackjack.py:48-160), the toy_text env that is not a P table ------------
 *    Observation = (player total, dealer's first card, usable ace) as three int64 columns obs[3][N] (Tuple(Discrete(32),
 *    Discrete(11), Discrete(2)) batched: three MultiDiscrete arrays); actions int64 {0 stick, 1 hit}; reward float64.
 *    Cards, deck = [1..10, 10, 10, 10] (:14-19), from the Philox draw stream (round-5 contract): key = env seed, ctr = (t_lo, t_hi,
 *    call, 5 << 28); every word yields TWO cards, the first two base-13 digits of word / 2^32 (d0 = (word * 13) >> 32, d1 = ((word * 13
 *    mod 2^32) * 13) >> 32, card = deck[d]: jointly uniform up to 169 / 2^32 = 4e-8).  The eight cards of call 0 have fixed roles —
 *    cards 0..3 (words x, y): the hit card resp. the dealer's first four draws of a stick; cards 4, 5 (word z): the next episode's dealer
 *    hand; cards 6, 7 (word w): the next player hand — and the dealer's draw j >= 4 is card (j + 4) & 7 of call (j + 4) >> 3: one Philox
 *    call per step, straight-line code.  Explicit reset: key = env seed, ctr = (t_lo, t_hi, r, 2 << 28), r = ordinal of the reset call,
 *    one card per word, deck[(word * 13) >> 32] (words x, y dealer; z, w player).  Sampled actions: the Discrete(2) bit stream of the
 *    RNG contract above (stream id 6).  cards_*: optional injected draws, int8 [N][MXV_BJ_MAX_DRAWS] per step (resp. [N][4] for a
 *    reset) in the reference's consumption order (the hit card or the dealer's cards, then on termination the new dealer hand, then the
 *    new player hand) — the values np_random.choice(deck) returned, for bit-exact replays. ----------------------------------------- */
#define MXV_BJ_MAX_DRAWS 24
typedef struct mxv_bj mxv_bj;
typedef struct mxv_bj_config {
    int32_t device;
    int32_t natural;           /* BlackjackEnv(natural=...): a winning natural pays 1.5 (ignored when sab) */
    int32_t sab;               /* BlackjackEnv(sab=...): Sutton & Barto rules (Blackjack-v1 registers sab=True) */
    int32_t max_episode_steps; /* <= 0: none (Blackjack-v1 has no TimeLimit) */
    int64_t num_envs;
    int64_t env_offset;
    uint64_t seed;
    uint64_t action_seed;
} mxv_bj_config;
int mxv_bj_create(const mxv_bj_config *cfg, mxv_bj **out);
int mxv_bj_destroy(mxv_bj *h);
const char *mxv_bj_last_error(const mxv_bj *h);
int mxv_bj_seed(mxv_bj *h, uint64_t base_seed, const uint64_t *per_env_seeds_host, uint64_t action_seed);
int mxv_bj_reset(mxv_bj *h, const uint8_t *mask_dev, const int8_t *cards_dev, int64_t *obs_dev);
int mxv_bj_step(mxv_bj *h, const int64_t *actions_dev, const int8_t *cards_dev, int64_t *obs_dev, double *reward_dev,
                uint8_t *terminated_dev, uint8_t *truncated_dev, int64_t *final_obs_dev);
/* K steps in one launch; actions from actions_tape_dev int64 [K][N], or sampled (NULL; recorded in actions_out_dev if given);
 * per_step != 0: outputs are [K][...] trajectories (obs [K][3][N]). */
int mxv_bj_rollout(mxv_bj *h, int32_t K, int32_t per_step, const int64_t *actions_tape_dev, int64_t *actions_out_dev,
                   int64_t *obs_dev, double *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, int64_t *final_obs_dev);
/* The same with the contract's 4-byte scalars: int32 observations / actions, float32 rewards — 22 B stored per
 * env-step instead of 42. */
int mxv_bj_rollout_compact(mxv_bj *h, int32_t K, int32_t per_step, const int64_t *actions_tape_dev, int32_t *actions_out_dev,
                           int32_t *obs_dev, float *reward_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, int32_t *final_obs_dev);
int mxv_bj_reset_host(mxv_bj *h, const int8_t *cards_host, int64_t *obs_host);
int mxv_bj_step_host(mxv_bj *h, const int64_t *actions_host, const int8_t *cards_host, int64_t *obs_host, double *reward_host,
                     uint8_t *terminated_host, uint8_t *truncated_host, int64_t *final_obs_host);
/* packed hands (see mxv_bj.hip) + TimeLimit counters, int32 [N] each; set_state also restores the step index / reset ordinal */
int mxv_bj_get_state(mxv_bj *h, int32_t *state_host, int32_t *elapsed_host);
int mxv_bj_set_state(mxv_bj *h, const int32_t *state_host, const int32_t *elapsed_host, uint64_t t, uint32_t r);
/* step index / reset ordinal of the draw streams (checkpointing: what mxv_bj_set_state takes back) */
int mxv_bj_get_counters(mxv_bj *h, uint64_t *t, uint32_t *r);
int mxv_bj_set_device_clock(mxv_bj *h, int32_t on);   /* as mxv_tab_set_device_clock: mxv_bj_step / mxv_bj_rollout recordable in a caller's hipGraph */
int mxv_bj_sync(mxv_bj *h);
int mxv_bj_set_stream(mxv_bj *h, void *stream);

/* -- collectives of a sharded vector env ------------------------------------------------------------------
 *    One logical vector env of N_total envs = `world` handles, one per GPU / process, rank r holding the contiguous global
 *    index range [r * N, (r+1) * N) (mxv_config.env_offset = r * N).  Env instances never interact (gym/vector/vector_env.py:
 *    13-16), so stepping needs no collective; the one exchange is the concatenation np.stack performs in the reference
 *    (gym/vector/sync_vector_env.py:159-169; AsyncVectorEnv gathers its workers' results the same way, async_vector_env.py:
 *    319-346): an all-gather of the shards' step outputs in rank order, here over RCCL / xGMI (librccl.so is opened with
 *    dlopen at mxv_comm_init: libmxv.so has no link-time dependency on it).
 *    Bootstrap like NCCL: one rank calls mxv_comm_unique_id and ships the MXV_COMM_ID_BYTES bytes to the others by any means
 *    (MPI, a file, torch.distributed's store ...); then every rank calls mxv_comm_init on its handle. ----------------------- */
#define MXV_COMM_ID_BYTES 128
int mxv_comm_unique_id(void *id_out);
int mxv_comm_init(mxv_handle *h, int32_t rank, int32_t world, const void *unique_id);
int mxv_comm_destroy(mxv_handle *h);
/* Asynchronous all-gather of this shard's outputs (obs float32 [N][O], reward in the handle's reward dtype [N], terminated /
 * truncated uint8 [N]; device pointers, e.g. the last slices of a rollout chunk's trajectory tensors) into [world][...]
 * device buffers = the full (N_total, ...) tensors in global env order.  The four gathers are issued as ONE grouped RCCL
 * launch on the communicator's own high-priority stream, ordered after everything launched so far on the handle's stream;
 * the call returns immediately and later launches on the handle's stream (the next rollout chunk) overlap it.  Any
 * send/receive pair may be NULL (skipped).  The send buffers must stay untouched until the gather has completed. */
int mxv_allgather_outputs(mxv_handle *h, const float *obs_dev, const void *reward_dev, const uint8_t *terminated_dev,
                          const uint8_t *truncated_dev, float *all_obs_dev, void *all_reward_dev, uint8_t *all_terminated_dev,
                          uint8_t *all_truncated_dev);
/* Wait for the last gather (age 0) or the one before it (age 1: what a caller that alternates between two snapshot buffers
 * needs before it lets a rollout overwrite the older one — the younger gather keeps overlapping).  host_sync == 0: the handle's
 * stream waits on the GPU, the host does not block; host_sync != 0: block the host until those gathered tensors are complete. */
int mxv_allgather_wait(mxv_handle *h, int32_t age, int32_t host_sync);
/* the hipStream_t the gathers run on (NULL before mxv_comm_init) */
int mxv_comm_stream(mxv_handle *h, void **stream);

/* -- diagnostics ------------------------------------------------------------------------------------------------------------------
 * What this GPU sustains for the store pattern of the fused CartPole rollout with the physics removed (one wave per workgroup, two
 * envs per lane, XCD-aware tiles; obs float32 [K][N][4], reward float64 [K][N], actions int64 [K][N], two flag bytes [K][N]: 34 B
 * per env-step): microseconds per vector step, averaged over `launches` K-step launches into the caller's [K][num_envs] buffers
 * (contents destroyed).  MI355X boxes differ by 20 % on this pattern; bench.py prints the figure next to the
 * kernel's own time so that a number can be read against the box it was taken on.  Synchronises. */
int mxv_write_probe(int32_t device, int64_t num_envs, int32_t K, int32_t launches, float *obs_dev, double *reward_dev, int64_t *actions_dev,
                    uint8_t *terminated_dev, uint8_t *truncated_dev, double *us_per_step);

/* The same for any env kind and output dtypes (flags: MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32): observation rows of that kind's width,
 * envs per lane as its fused rollout runs them, Box actions float32.  mxv_write_probe_env(MXV_CARTPOLE, 0, ...) is mxv_write_probe. */
int mxv_write_probe_env(int32_t device, int32_t env_id, int32_t flags, int64_t num_envs, int32_t K, int32_t launches, float *obs_dev,
                        void *reward_dev, void *actions_dev, uint8_t *terminated_dev, uint8_t *truncated_dev, double *us_per_step);

/* -- placed device memory for trajectory tensors ------------------------------------------------------------------------------------
 *    The fused rollout's outputs are a few long, parallel store streams.  On the MI355X a 16-byte-per-lane stream (observations) and
 *    an 8-byte-per-lane stream (rewards, actions) written concurrently run 10-12 % slower when the PHYSICAL memory behind them lies in
 *    the same CLASS of HBM regions — the classes are three contiguous thirds of the physical address space (3 x 96 GB: what the three
 *    ranks of a 12-high HBM3E stack would give; profiles/r3/r3c_hbm_class_map_whole_device.jsonl) —; a whole CartPole trajectory launch
 *    runs 5.4 / 5.7 / 6.4 us per 2^20-env step with none / one / both of {rewards, actions} in the observations' class (DESIGN.md §6,
 *    profiles/r3/r3a_*).  A fresh process is handed the first third for its first ~90 GiB, so hipMalloc'ed tensors all share a class unless
 *    earlier activity scrambled the driver's free lists — the "placement lottery" of rounds 1-2.  This call builds the tensors from
 *    256-MiB physical chunks (hipMemCreate) whose class it MEASURES (two concurrent streams against a reference chunk of each class
 *    seen so far) and maps chunks of one class under the tensors of one group and chunks of the other classes under the other group
 *    (group -1: whatever is left), each tensor contiguous in a fresh virtual range.
 *    Transient physical memory: chunks up to 2x the request; in addition, while only ONE class has been seen, unmapped spacer
 *    allocations (4 GiB each) that make the next chunk come from further along in physical memory — up to half of the device's free
 *    memory (at most 112 GiB; never into the last 16 GiB), released before the call returns; MXV_PLACED_NO_JUMP forbids the spacers
 *    (then a process that sits deep inside one class gets best effort: info.balanced = 0).  0.2-1.5 s.
 *    Sets below MXV_PLACED_MIN_BYTES (2 GiB: the real kernel runs 4-10 % slower on memory mapped through this API than on hipMalloc'ed
 *    memory, which the 8 % a 2^17-env shard of 1 GiB gains from separated classes does not win back — such sets are better served by
 *    ordinary allocations SORTED by class with mxv_hbm_pair_probe, what gym_amd/placement.py does from 1 GiB on), sets with an empty
 *    group and MXV_PLACED_PLAIN take ordinary hipMalloc allocations (info.placed = 0).
 *    bytes[i] > 0, group[i] in {-1, 0, 1}; ptrs_out[i] receives tensor i's device address (contents uninitialised).  mxv_placed_free
 *    releases the physical memory; the virtual ranges are NOT returned to the runtime (this runtime keeps stale translations for an
 *    address that is mapped a second time), i.e. every call consumes a little virtual address space for the life of the process. */
typedef struct mxv_placed mxv_placed;
#define MXV_PLACED_CHUNK_BYTES ((size_t)256 << 20)
#define MXV_PLACED_MIN_BYTES ((size_t)2 << 30)
enum { MXV_PLACED_PLAIN = 1, MXV_PLACED_NO_JUMP = 2 };
typedef struct mxv_placed_info {
    int32_t placed;            /* 1: chunks placed by class; 0: ordinary allocations */
    int32_t balanced;          /* 1: the two groups share no class */
    int32_t chunks_created;    /* physical chunks created (and classified) in total */
    int32_t chunks_kept;
    int32_t classes_seen;
    int32_t class_chunks[4];   /* chunks held of each class when the search ended (class 0 = the class of the first chunk) */
    int32_t solo_group;        /* the group that sits alone on class solo_class; the other group takes the other classes */
    int32_t solo_class;
    int32_t stop_reason;       /* why the search ended: 0 balanced, 1 chunk cap, 2 jump budget, 3 spacer allocation failed, 4 chunk allocation failed */
    double same_class_us;      /* the two-stream probe window, us per 2^20-lane step, both streams in one class ... */
    double different_class_us; /* ... and in different classes (0 if never seen) */
    double seconds;            /* wall time of the call */
    size_t requested_bytes, held_bytes, peak_bytes, jumped_bytes; /* peak: chunks + spacers at the worst moment; jumped: spacers */
} mxv_placed_info;
/* The measurement underneath: one 16-step window of two concurrent store streams of the rollout's launch shape (2^20 lanes), a 16-B/lane
 * stream over the 256 MiB at wide_dev and an 8-B/lane stream over the 128 MiB at narrow_dev (contents destroyed), us per step, best of
 * three timings of `launches` launches.  The same-class time of a box is ~4.2-4.4 us, a different-class pair runs at 0.89-0.91 of it:
 * compare against a pair known to share a class (two halves of one allocation), timed next to it. */
int mxv_hbm_pair_probe(int32_t device, void *wide_dev, void *narrow_dev, int32_t launches, double *us_per_step);
int mxv_placed_alloc(int32_t device, int32_t count, const size_t *bytes, const int32_t *group, int32_t flags, void **ptrs_out,
                     mxv_placed **out);
int mxv_placed_free(mxv_placed *p);
int mxv_placed_info_get(const mxv_placed *p, mxv_placed_info *out);
const char *mxv_placed_last_error(const mxv_placed *p); /* p may be NULL: last failed mxv_placed_alloc on this thread */

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
#endif /* MXV_H */
