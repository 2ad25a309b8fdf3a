/*
 * oracle/tabular.c — CPU restatement of the reference's tabular toy_text environments under SyncVectorEnv (SURVEY.md
 * §8f-4).  TEST INFRASTRUCTURE ONLY: the checker of the mxv_tab_* kernels, never a fallback for them.
 *
 * Follows /root/reference/gym/envs/toy_text:
 *   utils.py:4-8              categorical_sample(prob_n, rng) = argmax(cumsum(prob_n) > rng.random())
 *   frozen_lake.py:247-270    step: transitions = P[s][a]; i = categorical_sample(probs); p, s, r, t = transitions[i]
 *   taxi.py:254-278, cliffwalking.py:148-166  (the same step/reset over their own P tables)
 * and gym/wrappers/time_limit.py:39-68 (elapsed/truncated), gym/vector/sync_vector_env.py:135-169 (autoreset: the
 * returned observation/info of a finished sub-env are its reset()'s, the terminal ones go to final_observation/final_info).
 * The MDP arrives as the dense tables of include/mxv.h (cum_prob = np.cumsum of each transition list, padding -1).
 * The uniforms are either injected (the values the reference's own np_random.random() returned — how the goldens are
 * replayed bit for bit) or drawn from the engine's Philox contract (include/mxv.h: transition stream 3 — counter t >> 1, words
 * (x, y) for even steps and (z, w) for odd ones —, reset stream 2, action stream 1).
 */
#include <stdint.h>

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

static double tab_u01(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

static uint32_t tab_action_word(uint64_t action_seed, uint64_t t, uint64_t env) {
    uint32_t ctr[4], key[2], out[4];
    uint64_t g = env >> 2;
    ctr[0] = (uint32_t)g; ctr[1] = (uint32_t)(g >> 32);
    ctr[2] = (uint32_t)t; ctr[3] = ((uint32_t)(t >> 32) & 0x0fffffffu) | (1u << 28);
    key[0] = (uint32_t)action_seed; key[1] = (uint32_t)(action_seed >> 32);
    orc_philox4x32_10(ctr, key, out);
    return out[env & 3];
}

static void tab_stream_words(uint64_t seed, uint64_t t, uint32_t r, uint32_t stream, uint32_t out[4]) {
    uint32_t ctr[4], key[2];
    ctr[0] = (uint32_t)t; ctr[1] = (uint32_t)(t >> 32); ctr[2] = r; ctr[3] = stream << 28;
    key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    orc_philox4x32_10(ctr, key, out);
}

/* np.argmax(csprob_n > u): index of the first True, 0 when none (padding entries are -1: never True) */
static int categorical(const double *cum, int len, double u) {
    for (int i = 0; i < len; ++i)
        if (cum[i] > u) return i;
    return 0;
}

void orc_tab_reset(int S, const double *init_cum, int64_t n, uint64_t env0, const uint64_t *seeds, uint64_t base_seed,
                   uint64_t t, uint32_t r, const uint8_t *mask, int32_t *state, int32_t *elapsed, int64_t *obs) {
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) {
            obs[i] = state[i];
            continue;
        }
        uint32_t w[4];
        tab_stream_words(seeds ? seeds[i] : base_seed + env0 + (uint64_t)i, t, r, 2u, w);
        state[i] = categorical(init_cum, S, tab_u01(w[0]));
        elapsed[i] = 0;
        obs[i] = state[i];
    }
}

/* One vector step.  actions NULL -> sampled; uniforms NULL -> Philox, else double[2][n].  Returns #invalid actions. */
int64_t orc_tab_step(int S, int A, int M, const double *cum, const double *prob, const int32_t *next, const double *reward,
                     const uint8_t *term_tab, const double *init_cum, int64_t n, uint64_t env0, const uint64_t *seeds,
                     uint64_t base_seed, uint64_t action_seed, uint64_t t, int max_steps, const int64_t *actions,
                     const double *uniforms, int32_t *state, int32_t *elapsed, int64_t *actions_out, int64_t *obs,
                     double *rew, uint8_t *term, uint8_t *trunc, double *prob_out, int64_t *final_obs, double *final_prob,
                     uint8_t *final_mask) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t ge = env0 + (uint64_t)i;
        int64_t a;
        if (actions) {
            a = actions[i];
            if (a < 0 || a >= A) { bad++; continue; }
        } else {
            a = (int64_t)(((uint64_t)tab_action_word(action_seed, t, ge) * (uint64_t)A) >> 32);
        }
        if (actions_out) actions_out[i] = a;
        double u_step, u_reset;
        if (uniforms) {
            u_step = uniforms[i];
            u_reset = uniforms[n + i];
        } else {
            uint32_t w[4];
            tab_stream_words(seeds ? seeds[i] : base_seed + ge, t >> 1, 0u, 3u, w); /* one call serves steps 2b, 2b+1 */
            u_step = tab_u01(w[(t & 1) ? 2 : 0]);
            u_reset = tab_u01(w[(t & 1) ? 3 : 1]);
        }
        const int base = (state[i] * A + (int)a) * M;
        const int j = base + categorical(cum + base, M, u_step);   /* frozen_lake.py:248-250 */
        double p = prob[j];
        const int32_t ns = next[j];
        const int te = term_tab[j] != 0;
        rew[i] = reward[j];
        elapsed[i] += 1;                                            /* time_limit.py:50-53 */
        const int tr = max_steps > 0 && elapsed[i] >= max_steps;
        state[i] = ns;
        final_mask[i] = 0;
        if (te || tr) {                                             /* sync_vector_env.py:152-156 */
            final_obs[i] = ns;
            final_prob[i] = p;
            final_mask[i] = 1;
            state[i] = categorical(init_cum, S, u_reset);           /* reset(): frozen_lake.py:264-265 */
            elapsed[i] = 0;
            p = 1.0;
        }
        obs[i] = state[i];
        term[i] = (uint8_t)te;
        trunc[i] = (uint8_t)tr;
        prob_out[i] = p;
    }
    return bad;
}
