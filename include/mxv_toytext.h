/* mxv_toytext.h — the toy_text engines: tabular MDPs (FrozenLake / Taxi / CliffWalking) and Blackjack-v1 (API level 2, Blackjack draw contract: level 5).
 * Part of the C ABI of libmxv.so (see mxv.h for the engine's handle, status codes, RNG and numerical contracts).  Including mxv.h
 * includes this file. */
#ifndef MXV_TOYTEXT_H
#define MXV_TOYTEXT_H

#include "mxv.h"

#ifdef __cplusplus
extern "C" {
#endif

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
/* -- episode statistics (API level 6): gym.wrappers.RecordEpisodeStatistics (gym/wrappers/record_episode_statistics.py:96-151) fused into
 * the step and trajectory kernels, as mxv_episode_stats does for the classic-control engine: enable -> a float32 running return per env
 * (episode length is the TimeLimit counter), zeroed for every env an explicit reset covers.  set_episode_outputs: device arrays [N] /
 * [K][N] for the device-pointer calls (entries are MEANINGFUL only where that step's terminated | truncated is set: a wave in which an
 * episode ended stores its 64 entries as whole lines, zeros where none ended; other waves store nothing); episode_stats_host:
 * the returns / lengths of the episodes that ended in the LAST host step (valid where its terminated | truncated is set) and the running
 * returns of all envs; any pointer may be NULL. */
int mxv_tab_episode_stats(mxv_tab *h, int32_t enable);
int mxv_tab_set_episode_outputs(mxv_tab *h, float *ep_return_dev, int32_t *ep_length_dev);
int mxv_tab_episode_stats_host(mxv_tab *h, float *ep_return_host, int32_t *ep_length_host, float *running_return_host);
/* checkpoint restore of the running returns read with episode_stats_host(running_return_host): float32 [N] */
int mxv_tab_set_running_returns(mxv_tab *h, const float *running_return_host);
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
/* episode statistics of the Blackjack engine: as mxv_tab_episode_stats / _set_episode_outputs / _episode_stats_host above */
int mxv_bj_episode_stats(mxv_bj *h, int32_t enable);
int mxv_bj_set_episode_outputs(mxv_bj *h, float *ep_return_dev, int32_t *ep_length_dev);
int mxv_bj_episode_stats_host(mxv_bj *h, float *ep_return_host, int32_t *ep_length_host, float *running_return_host);
int mxv_bj_set_running_returns(mxv_bj *h, const float *running_return_host);
/* packed hands (see mxv_bj.hip) + TimeLimit counters, int32 [N] each; set_state also restores the step index / reset ordinal */
int mxv_bj_get_state(mxv_bj *h, int32_t *state_host, int32_t *elapsed_host);
int mxv_bj_set_state(mxv_bj *h, const int32_t *state_host, const int32_t *elapsed_host, uint64_t t, uint32_t r);
/* step index / reset ordinal of the draw streams (checkpointing: what mxv_bj_set_state takes back) */
int mxv_bj_get_counters(mxv_bj *h, uint64_t *t, uint32_t *r);
int mxv_bj_set_device_clock(mxv_bj *h, int32_t on);   /* as mxv_tab_set_device_clock: mxv_bj_step / mxv_bj_rollout recordable in a caller's hipGraph */
int mxv_bj_sync(mxv_bj *h);
int mxv_bj_set_stream(mxv_bj *h, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MXV_TOYTEXT_H */
