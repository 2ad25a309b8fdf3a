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

/* ------------------------------------------------------------------------------------------------------------------------
 * Blackjack-v1 — gym/envs/toy_text/blackjack.py.  Hands are kept as card lists exactly like the reference (:17-45); the
 * cards come either from the caller (the values np_random.choice(deck) returned, :18, in the reference's consumption order) or
 * from the engine's Philox draw stream (include/mxv.h, round-5 contract): key = env seed, ctr = (t_lo, t_hi, call, 5 << 28); every
 * word yields TWO cards, the first two base-13 digits of word / 2^32 (d0 = (word * 13) >> 32, d1 = ((word * 13 mod 2^32) * 13) >> 32,
 * card = deck[d]); the eight cards of call 0 have fixed roles — cards 0..3 (words x, y): the hit card resp. the dealer's first four
 * draws; cards 4, 5 (word z): the next episode's dealer hand; cards 6, 7 (word w): the next player hand — and the dealer's draw
 * j >= 4 is card (j + 4) & 7 of call (j + 4) >> 3.  Sampled actions: the engine's Discrete(2) bit stream (stream 6: one call per
 * 32 steps, action = bit t & 31 of the env's word), as for CartPole.
 * ---------------------------------------------------------------------------------------------------------------------- */
#define BJ_MAX_HAND 32
typedef struct { int c[BJ_MAX_HAND]; int n; } bj_hand;
static int bj_sum(const bj_hand *h) { int s = 0; for (int i = 0; i < h->n; ++i) s += h->c[i]; return s; }
static int bj_has_ace(const bj_hand *h) { for (int i = 0; i < h->n; ++i) if (h->c[i] == 1) return 1; return 0; }
static int bj_usable(const bj_hand *h) { return bj_has_ace(h) && bj_sum(h) + 10 <= 21; }          /* :26-27 */
static int bj_total(const bj_hand *h) { return bj_usable(h) ? bj_sum(h) + 10 : bj_sum(h); }        /* :30-33 */
static int bj_score(const bj_hand *h) { return bj_total(h) > 21 ? 0 : bj_total(h); }               /* :36-41 */
static int bj_natural(const bj_hand *h) {                                                          /* sorted(hand) == [1, 10] */
    return h->n == 2 && ((h->c[0] == 1 && h->c[1] == 10) || (h->c[0] == 10 && h->c[1] == 1));
}
static const int BJ_DECK[13] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 10};

/* card g (0, 1, 2, ...) of the step's draw stream: digit g & 1 of word (g >> 1) & 3 of call g >> 3 */
static int bj_stream_card(uint64_t seed, uint64_t t, int g) {
    uint32_t w[4];
    tab_stream_words(seed, t, (uint32_t)(g >> 3), 5u, w);
    const uint64_t p = (uint64_t)w[(g >> 1) & 3] * 13u;
    const uint32_t d0 = (uint32_t)(p >> 32), d1 = (uint32_t)(((uint64_t)(uint32_t)p * 13u) >> 32);
    return BJ_DECK[(g & 1) ? d1 : d0];
}
typedef struct { const int8_t *inj; uint64_t seed, t; int cursor, draws; } bj_src;
/* the step's next draw (the hit card / a dealer card) */
static int bj_draw(bj_src *s) {
    const int j = s->draws++;
    if (s->inj) return s->inj[s->cursor++];
    return bj_stream_card(s->seed, s->t, j < 4 ? j : j + 4);
}
/* card k (0..3) of the hands dealt after the step: dealer's two, then the player's two */
static int bj_deal(bj_src *s, int k) {
    if (s->inj) return s->inj[s->cursor++];
    return bj_stream_card(s->seed, s->t, 4 + k);
}
static uint32_t tab_action_bits_word(uint64_t action_seed, uint64_t t, uint64_t env) {
    uint32_t ctr[4], key[2], out[4];
    const uint64_t g = env >> 2, b = t >> 5;
    ctr[0] = (uint32_t)g; ctr[1] = (uint32_t)(g >> 32);
    ctr[2] = (uint32_t)b; ctr[3] = ((uint32_t)(b >> 32) & 0x0fffffffu) | (6u << 28);
    key[0] = (uint32_t)action_seed; key[1] = (uint32_t)(action_seed >> 32);
    orc_philox4x32_10(ctr, key, out);
    return out[env & 3];
}

/* test hook (tests/test_blackjack_oracle.py: the distribution of the card map): the first `count` cards of the draw stream of n envs
 * (seeds base_seed + i) at step t, int8 out[n][count] */
void orc_bj_stream_cards(int64_t n, uint64_t base_seed, uint64_t t, int count, int8_t *out) {
    for (int64_t i = 0; i < n; ++i)
        for (int g = 0; g < count; ++g) out[i * count + g] = (int8_t)bj_stream_card(base_seed + (uint64_t)i, t, g);
}

/* state per env: dealer and player card lists (caller-owned arrays of bj_hand-compatible layout: int[33] each = 32 cards + n) */
void orc_bj_reset(int64_t n, uint64_t env0, const uint64_t *seeds, uint64_t base_seed, uint64_t t, uint32_t r,
                  const int8_t *cards, int32_t *dealer, int32_t *player, int32_t *elapsed, int64_t *obs) {
    for (int64_t i = 0; i < n; ++i) {
        bj_hand *d = (bj_hand *)(dealer + i * (BJ_MAX_HAND + 1)), *p = (bj_hand *)(player + i * (BJ_MAX_HAND + 1));
        int c[4];
        if (cards) {
            for (int k = 0; k < 4; ++k) c[k] = cards[i * 4 + k];
        } else {
            uint32_t w[4];
            tab_stream_words(seeds ? seeds[i] : base_seed + env0 + (uint64_t)i, t, r, 2u, w);
            for (int k = 0; k < 4; ++k) c[k] = BJ_DECK[(int)(((uint64_t)w[k] * 13u) >> 32)];
        }
        d->c[0] = c[0]; d->c[1] = c[1]; d->n = 2;      /* :157 dealer first */
        p->c[0] = c[2]; p->c[1] = c[3]; p->n = 2;      /* :158 */
        elapsed[i] = 0;
        obs[i] = bj_total(p); obs[n + i] = d->c[0]; obs[2 * n + i] = bj_usable(p);
    }
}

int64_t orc_bj_step(int64_t n, uint64_t env0, const uint64_t *seeds, uint64_t base_seed, uint64_t action_seed, uint64_t t,
                    int natural, int sab, int max_steps, const int64_t *actions, const int8_t *cards, int max_draws,
                    int32_t *dealer, int32_t *player, int32_t *elapsed, int64_t *actions_out, int64_t *obs, double *rew,
                    uint8_t *term, uint8_t *trunc, int64_t *final_obs, uint8_t *final_mask) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        bj_hand *d = (bj_hand *)(dealer + i * (BJ_MAX_HAND + 1)), *p = (bj_hand *)(player + i * (BJ_MAX_HAND + 1));
        const uint64_t ge = env0 + (uint64_t)i;
        int64_t a;
        if (actions) {
            a = actions[i];
            if (a < 0 || a > 1) { bad++; continue; }
        } else {
            a = (int64_t)((tab_action_bits_word(action_seed, t, ge) >> (uint32_t)(t & 31u)) & 1u);
        }
        if (actions_out) actions_out[i] = a;
        bj_src src = {cards ? cards + i * max_draws : 0, seeds ? seeds[i] : base_seed + ge, t, 0, 0};
        int te;
        double r;
        if (a) {                                                   /* hit :123-130 */
            p->c[p->n++] = bj_draw(&src);
            te = bj_total(p) > 21;
            r = te ? -1.0 : 0.0;
        } else {                                                   /* stick :131-146 */
            te = 1;
            while (bj_total(d) < 17) d->c[d->n++] = bj_draw(&src);
            const int ps = bj_score(p), ds = bj_score(d);
            r = (double)(ps > ds) - (double)(ps < ds);
            if (sab && bj_natural(p) && !bj_natural(d)) r = 1.0;
            else if (!sab && natural && bj_natural(p) && r == 1.0) r = 1.5;
        }
        elapsed[i] += 1;
        const int tr = max_steps > 0 && elapsed[i] >= max_steps;
        final_mask[i] = 0;
        if (te || tr) {
            final_obs[i] = bj_total(p); final_obs[n + i] = d->c[0]; final_obs[2 * n + i] = bj_usable(p);
            final_mask[i] = 1;
            d->c[0] = bj_deal(&src, 0); d->c[1] = bj_deal(&src, 1); d->n = 2;
            p->c[0] = bj_deal(&src, 2); p->c[1] = bj_deal(&src, 3); p->n = 2;
            elapsed[i] = 0;
        }
        obs[i] = bj_total(p); obs[n + i] = d->c[0]; obs[2 * n + i] = bj_usable(p);
        rew[i] = r; term[i] = (uint8_t)te; trunc[i] = (uint8_t)tr;
    }
    return bad;
}
