/* mxv_norm.h — running normalisation: gym.wrappers.NormalizeObservation / NormalizeReward (API level 2).
 * Part of the C ABI of libmxv.so (see mxv.h for the engine's handle, status codes, RNG and numerical contracts).  Including mxv.h
 * includes this file. */
#ifndef MXV_NORM_H
#define MXV_NORM_H

#include "mxv.h"

#ifdef __cplusplus
extern "C" {
#endif

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

/* -- the PER-SUB-ENV form (API level 6): `gym.vector.make(id, n, wrappers=[NormalizeObservation, NormalizeReward])` puts the wrappers
 *    around EVERY sub-env (gym/vector/__init__.py:56-65), so each sub-env owns a RunningMeanStd updated with batches of ONE row
 *    (normalize.py:17-47 with batch_mean = the row, batch_var = 0, batch_count = 1) — a different normalisation from the vector-level
 *    wrappers above, with no cross-env reduction.  One mxv_subnorm = num_envs RunningMeanStd objects of shape (dim,) (fp64 mean / var /
 *    count per sub-env, device resident) plus, for rewards (dim == 1), every sub-env's discounted return.  Works on K consecutive
 *    batches per call like mxv_norm ([K][num_envs][dim]; K = 1 for a single step()); the statistics stay in registers across the K
 *    steps.  Arithmetic and order are the reference's; results are bit-identical to the NumPy wrappers' on the same inputs.
 *    dim in {1, 2, 3, 4, 6}.  Asynchronous on `stream`. ------------------------------------------------------------------------- */
typedef struct mxv_subnorm mxv_subnorm;
int mxv_subnorm_create(int32_t device, int32_t dim, int64_t num_envs, void *stream, mxv_subnorm **out);
int mxv_subnorm_destroy(mxv_subnorm *nm);
const char *mxv_subnorm_last_error(const mxv_subnorm *nm); /* nm may be NULL: last failed mxv_subnorm_create on this thread */
int mxv_subnorm_set_stream(mxv_subnorm *nm, void *stream);
/* every sub-env's obs_rms.mean / .var (double[num_envs][dim]), .count (double[num_envs]) and NormalizeReward.returns
 * (double[num_envs]); any pointer may be NULL (set_state: mean, var and count are required).  Synchronises. */
int mxv_subnorm_get_state(mxv_subnorm *nm, double *mean_host, double *var_host, double *count_host, double *returns_host);
int mxv_subnorm_set_state(mxv_subnorm *nm, const double *mean_host, const double *var_host, const double *count_host,
                          const double *returns_host);
/* NormalizeObservation around every sub-env (normalize.py:72-93 under sync_vector_env.py:142-156).  x_dev float32 [K][num_envs][dim]:
 * the batched observations of K steps (the post-autoreset row where an episode ended); final_dev float32 [K][num_envs][dim]: the
 * terminal observations (rows valid where terminated | truncated), terminated_dev / truncated_dev uint8 [K][num_envs].  Per sub-env and
 * step: where the episode ended the terminal row is folded in and normalised first (-> final_y_dev float64 [K][num_envs][dim], rows of
 * finished sub-envs only; may be NULL), then the batched row (-> y_dev, float32 when out_f32 != 0 — the dtype of the reference's
 * batched observations — else float64; y_dev may alias x_dev only when out_f32 != 0).  reset(): final_dev = terminated_dev =
 * truncated_dev = NULL, K = 1. */
int mxv_subnorm_observations(mxv_subnorm *nm, int32_t K, const float *x_dev, const float *final_dev, const uint8_t *terminated_dev,
                             const uint8_t *truncated_dev, void *y_dev, int32_t out_f32, double *final_y_dev, double epsilon);
/* NormalizeReward around every sub-env (normalize.py:127-145): returns = returns * gamma + reward; return_rms.update(returns);
 * out = reward / sqrt(var + epsilon); returns = 0 where terminated | truncated.  reward / out float64 [K][num_envs] (float32 when
 * reward_f32 != 0); out_dev may alias reward_dev.  Needs dim == 1. */
int mxv_subnorm_rewards(mxv_subnorm *nm, int32_t K, const void *reward_dev, int32_t reward_f32, const uint8_t *terminated_dev,
                        const uint8_t *truncated_dev, void *out_dev, double gamma, double epsilon);

#ifdef __cplusplus
}
#endif
#endif /* MXV_NORM_H */
