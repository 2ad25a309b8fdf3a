/* tests/c_consumer/abi_consumer.c — a consumer of include/mxv.h written in C, with no Python and no ctypes mirror in between.
 *
 * The drop-in boundary of this engine is a C ABI.  Every other test reaches it through ctypes, i.e. through
 * gym_amd/_native.py's hand-written copy of the prototypes and struct layouts; this program takes both from the header itself:
 * the function pointers are declared with __typeof__(the header's prototype) and `mxv_config` is the header's struct.  It opens
 * libmxv.so and the CPU oracle (oracle/_build/liborc.so — test infrastructure, the checker), and for each of the five env kinds runs
 * what __graft_entry__.smoke() runs for CartPole: seeded reset, then vector steps with the oracle's Philox actions through
 * mxv_step_host, compared with orc_vec_step step by step — actions in range, masks and rewards exact (Pendulum's reward to 1e-12
 * relative), observations within 2 float32 ulps, reset states bit-exact, TimeLimit counters equal — resynchronising the oracle's
 * fp64 state from the device after every step so that last-bit libm differences do not accumulate.
 *
 *   gcc -std=gnu99 -Wall -Wextra -I include -o abi_consumer tests/c_consumer/abi_consumer.c -ldl -lm
 *   ./abi_consumer gym_amd/_lib/libmxv.so oracle/_build/liborc.so [num_envs] [steps]        (needs a HIP device)
 *   ./abi_consumer --symbols-only gym_amd/_lib/libmxv.so                                   (no device: resolves the symbols it uses)
 */
#include <dlfcn.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mxv.h"
#include "mxv_diag.h"   /* optional diagnostics: the consumer prints the layout of mxv_placed_info and mxv_launch_info too */

#define FN(name) static __typeof__(name) *p_##name
FN(mxv_version);
FN(mxv_env_dims);
FN(mxv_default_params);
FN(mxv_default_reset_bounds);
FN(mxv_create);
FN(mxv_destroy);
FN(mxv_last_error);
FN(mxv_reset_host);
FN(mxv_step_host);
FN(mxv_get_state);
FN(mxv_get_counters);
FN(mxv_get_episodes);
FN(mxv_bj_create);
FN(mxv_bj_destroy);
FN(mxv_bj_last_error);
FN(mxv_bj_reset_host);
FN(mxv_bj_step_host);
FN(mxv_bj_get_counters);

/* the oracle's entry points (oracle/classic_control.c) */
typedef void (*orc_vec_reset_t)(int, int64_t, uint64_t, const uint64_t *, uint64_t, uint32_t *, const double *, const uint8_t *, double *,
                                int32_t *, float *);
typedef int64_t (*orc_vec_step_t)(int, int64_t, uint64_t, const double *, int, int, const uint64_t *, uint64_t, uint64_t, uint32_t *,
                                  const double *, const int64_t *, const float *, double *, int32_t *, float *, double *, uint8_t *,
                                  uint8_t *, float *, uint8_t *);
typedef void (*orc_sample_actions_t)(int, int64_t, uint64_t, uint64_t, uint64_t, const double *, int64_t *, float *);
typedef void (*orc_bj_reset_t)(int64_t, uint64_t, const uint64_t *, uint64_t, uint64_t, uint32_t, const int8_t *, int32_t *, int32_t *,
                               int32_t *, int64_t *);
typedef int64_t (*orc_bj_step_t)(int64_t, uint64_t, const uint64_t *, uint64_t, uint64_t, uint64_t, int, int, int, const int64_t *,
                                 const int8_t *, int, int32_t *, int32_t *, int32_t *, int64_t *, int64_t *, double *, uint8_t *, uint8_t *,
                                 int64_t *, uint8_t *);
typedef void (*orc_default_params_t)(int, double *);
typedef void (*orc_default_reset_bounds_t)(int, double *);

static void *must(void *lib, const char *name) {
    void *p = dlsym(lib, name);
    if (!p) {
        fprintf(stderr, "missing symbol %s: %s\n", name, dlerror());
        exit(2);
    }
    return p;
}
#define LOAD(lib, name) p_##name = (__typeof__(p_##name))must(lib, #name)

static int64_t ulps32(float a, float b) {
    int32_t ia, ib;
    memcpy(&ia, &a, 4);
    memcpy(&ib, &b, 4);
    int64_t x = ia < 0 ? -(int64_t)(ia & 0x7fffffff) : ia, y = ib < 0 ? -(int64_t)(ib & 0x7fffffff) : ib;
    return x > y ? x - y : y - x;
}

static const char *NAMES[5] = {"CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0"};
static const int LIMITS[5] = {500, 200, 500, 200, 999}; /* gym/envs/__init__.py:11-50 */

int main(int argc, char **argv) {
    int symbols_only = argc > 1 && strcmp(argv[1], "--symbols-only") == 0;
    if (argc < (symbols_only ? 3 : 3)) {
        fprintf(stderr, "usage: %s [--symbols-only] libmxv.so [liborc.so [num_envs [steps]]]\n", argv[0]);
        return 2;
    }
    void *mxv = dlopen(argv[symbols_only ? 2 : 1], RTLD_NOW | RTLD_LOCAL);
    if (!mxv) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    LOAD(mxv, mxv_version);
    LOAD(mxv, mxv_env_dims);
    LOAD(mxv, mxv_default_params);
    LOAD(mxv, mxv_default_reset_bounds);
    LOAD(mxv, mxv_create);
    LOAD(mxv, mxv_destroy);
    LOAD(mxv, mxv_last_error);
    LOAD(mxv, mxv_reset_host);
    LOAD(mxv, mxv_step_host);
    LOAD(mxv, mxv_get_state);
    LOAD(mxv, mxv_get_counters);
    LOAD(mxv, mxv_get_episodes);
    LOAD(mxv, mxv_bj_create);
    LOAD(mxv, mxv_bj_destroy);
    LOAD(mxv, mxv_bj_last_error);
    LOAD(mxv, mxv_bj_reset_host);
    LOAD(mxv, mxv_bj_step_host);
    LOAD(mxv, mxv_bj_get_counters);
    printf("%s: sizeof(mxv_config) = %zu\n", p_mxv_version(), sizeof(mxv_config));
    /* static information needs no device */
    for (int env = 0; env < 5; ++env) {
        int32_t S = 0, O = 0, NA = -1;
        if (p_mxv_env_dims(env, &S, &O, &NA) != MXV_OK || S < 2 || S > 4 || O < 2 || O > 6) {
            fprintf(stderr, "mxv_env_dims(%d) -> S=%d O=%d NA=%d\n", env, S, O, NA);
            return 1;
        }
    }
    if (symbols_only) {
        /* every struct of the header as C lays it out: "layout <struct> <size> <field>=<offset> ..." (tests/test_c_consumer.py holds
         * gym_amd/_native.py's ctypes mirrors against these lines) */
#define OFF(T, f) printf(" " #f "=%zu", offsetof(T, f))
        printf("layout mxv_config %zu", sizeof(mxv_config));
        OFF(mxv_config, env_id); OFF(mxv_config, device); OFF(mxv_config, num_envs); OFF(mxv_config, env_offset);
        OFF(mxv_config, max_episode_steps); OFF(mxv_config, flags); OFF(mxv_config, seed); OFF(mxv_config, action_seed);
        printf("\nlayout mxv_tab_config %zu", sizeof(mxv_tab_config));
        OFF(mxv_tab_config, device); OFF(mxv_tab_config, num_states); OFF(mxv_tab_config, num_actions); OFF(mxv_tab_config, max_transitions);
        OFF(mxv_tab_config, num_envs); OFF(mxv_tab_config, env_offset); OFF(mxv_tab_config, max_episode_steps); OFF(mxv_tab_config, flags);
        OFF(mxv_tab_config, seed); OFF(mxv_tab_config, action_seed);
        printf("\nlayout mxv_bj_config %zu", sizeof(mxv_bj_config));
        OFF(mxv_bj_config, device); OFF(mxv_bj_config, natural); OFF(mxv_bj_config, sab); OFF(mxv_bj_config, max_episode_steps);
        OFF(mxv_bj_config, num_envs); OFF(mxv_bj_config, env_offset); OFF(mxv_bj_config, seed); OFF(mxv_bj_config, action_seed);
        printf("\nlayout mxv_placed_info %zu", sizeof(mxv_placed_info));
        OFF(mxv_placed_info, placed); OFF(mxv_placed_info, balanced); OFF(mxv_placed_info, chunks_created); OFF(mxv_placed_info, chunks_kept);
        OFF(mxv_placed_info, classes_seen); OFF(mxv_placed_info, class_chunks); OFF(mxv_placed_info, solo_group);
        OFF(mxv_placed_info, solo_class); OFF(mxv_placed_info, stop_reason); OFF(mxv_placed_info, same_class_us);
        OFF(mxv_placed_info, different_class_us); OFF(mxv_placed_info, seconds); OFF(mxv_placed_info, requested_bytes);
        OFF(mxv_placed_info, held_bytes); OFF(mxv_placed_info, peak_bytes); OFF(mxv_placed_info, jumped_bytes);
        printf("\nlayout mxv_launch_info %zu", sizeof(mxv_launch_info));
        OFF(mxv_launch_info, kernel); OFF(mxv_launch_info, env_id); OFF(mxv_launch_info, param_mode); OFF(mxv_launch_info, envs_per_lane);
        OFF(mxv_launch_info, safe); OFF(mxv_launch_info, out_mode); OFF(mxv_launch_info, tape); OFF(mxv_launch_info, steps);
        OFF(mxv_launch_info, grid); OFF(mxv_launch_info, block);
        printf("\nlayout mxv_step_outputs %zu\n", sizeof(mxv_step_outputs));
        printf("symbols ok\n");
        return 0;
    }
    if (argc < 3) return 2;
    void *orc = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!orc) {
        fprintf(stderr, "dlopen oracle: %s\n", dlerror());
        return 2;
    }
    orc_vec_reset_t orc_vec_reset = (orc_vec_reset_t)must(orc, "orc_vec_reset");
    orc_vec_step_t orc_vec_step = (orc_vec_step_t)must(orc, "orc_vec_step");
    orc_sample_actions_t orc_sample_actions = (orc_sample_actions_t)must(orc, "orc_sample_actions");
    orc_default_params_t orc_default_params = (orc_default_params_t)must(orc, "orc_default_params");
    orc_default_reset_bounds_t orc_default_reset_bounds = (orc_default_reset_bounds_t)must(orc, "orc_default_reset_bounds");

    const int64_t n = argc > 3 ? atoll(argv[3]) : 1000;
    const int steps = argc > 4 ? atoi(argv[4]) : 260;
    const uint64_t seed = 5, action_seed = 6, env0 = 4096;
    int failures = 0;
    for (int env = 0; env < 5; ++env) {
        int32_t S, O, NA;
        p_mxv_env_dims(env, &S, &O, &NA);
        mxv_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.env_id = env;
        cfg.device = 0;
        cfg.num_envs = n;
        cfg.env_offset = (int64_t)env0;
        cfg.max_episode_steps = LIMITS[env] < 40 ? LIMITS[env] : 37; /* short episodes: TimeLimit + autoreset inside the run */
        cfg.flags = 0;
        cfg.seed = seed;
        cfg.action_seed = action_seed;
        mxv_handle *h = NULL;
        if (p_mxv_create(&cfg, &h) != MXV_OK) {
            fprintf(stderr, "%s: mxv_create: %s\n", NAMES[env], p_mxv_last_error(NULL));
            return 1;
        }
        double P[MXV_MAX_PARAMS], Pm[MXV_MAX_PARAMS], bounds[2], bm[2];
        orc_default_params(env, P);
        orc_default_reset_bounds(env, bounds);
        p_mxv_default_params(env, Pm);
        p_mxv_default_reset_bounds(env, bm);
        if (bounds[0] != bm[0] || bounds[1] != bm[1]) {
            fprintf(stderr, "%s: default reset bounds differ\n", NAMES[env]);
            ++failures;
        }
        double *st = malloc(sizeof(double) * S * n), *dst = malloc(sizeof(double) * S * n), *rew = malloc(8 * n), *drew = malloc(8 * n);
        int32_t *el = calloc(n, 4), *del = malloc(4 * n);
        uint32_t *ep = calloc(n, 4), *dep = malloc(4 * n);
        float *obs = malloc(4 * O * n), *dobs = malloc(4 * O * n), *fin = malloc(4 * O * n), *dfin = malloc(4 * O * n), *af = malloc(4 * n);
        int64_t *ai = malloc(8 * n);
        uint8_t *te = malloc(n), *tr = malloc(n), *dte = malloc(n), *dtr = malloc(n), *fm = malloc(n);
        orc_vec_reset(env, n, env0, NULL, seed, ep, bounds, NULL, st, el, obs);
        if (p_mxv_reset_host(h, NULL, NULL, dobs) != MXV_OK) {
            fprintf(stderr, "%s: mxv_reset_host: %s\n", NAMES[env], p_mxv_last_error(h));
            return 1;
        }
        p_mxv_get_state(h, dst, del);
        int bad = memcmp(st, dst, sizeof(double) * S * n) != 0; /* reset states: bit-exact (Philox reset stream + u01 arithmetic) */
        long worst = 0, dones = 0;
        for (int t = 0; t < steps && !bad; ++t) {
            orc_sample_actions(env, n, env0, action_seed, (uint64_t)t, P, ai, af);
            const void *actions = NA > 0 ? (const void *)ai : (const void *)af;
            if (p_mxv_step_host(h, actions, dobs, drew, dte, dtr, dfin) != MXV_OK) {
                fprintf(stderr, "%s: mxv_step_host: %s\n", NAMES[env], p_mxv_last_error(h));
                return 1;
            }
            orc_vec_step(env, n, env0, P, cfg.max_episode_steps, 1, NULL, seed, (uint64_t)t, ep, bounds, ai, af, st, el, obs, rew, te, tr,
                         fin, fm);
            for (int64_t i = 0; i < n && !bad; ++i) {
                if (te[i] != dte[i] || tr[i] != dtr[i]) bad = 1;
                const double tol = env == 1 ? 1e-12 * fabs(rew[i]) + 1e-9 : 0.0; /* Pendulum: u**2 is libm powf in the reference */
                if (fabs(rew[i] - drew[i]) > tol) bad = 1;
                for (int k = 0; k < O; ++k) {
                    long u = (long)ulps32(obs[i * O + k], dobs[i * O + k]);
                    if (u > worst) worst = u;
                    if (u > 2) bad = 1;
                    if ((te[i] || tr[i]) && ulps32(fin[i * O + k], dfin[i * O + k]) > 2) bad = 1;
                }
                dones += te[i] | tr[i];
                if (bad) fprintf(stderr, "%s: mismatch at step %d env %lld\n", NAMES[env], t, (long long)i);
            }
            p_mxv_get_state(h, dst, del);
            if (memcmp(el, del, 4 * n) != 0) {
                fprintf(stderr, "%s: TimeLimit counters differ at step %d\n", NAMES[env], t);
                bad = 1;
            }
            memcpy(st, dst, sizeof(double) * S * n); /* keep last-bit differences from accumulating */
        }
        uint64_t tt = 0;
        uint32_t rr = 0;
        p_mxv_get_counters(h, &tt, &rr);
        p_mxv_get_episodes(h, dep);
        if (!bad && (tt != (uint64_t)steps || memcmp(ep, dep, 4 * n) != 0)) {
            fprintf(stderr, "%s: step index %llu (expected %d) or reset ordinals differ\n", NAMES[env], (unsigned long long)tt, steps);
            bad = 1;
        }
        printf("%-26s %s  envs=%lld steps=%d episodes_ended=%ld worst_obs_ulps=%ld\n", NAMES[env], bad ? "FAILED" : "ok", (long long)n, steps,
               dones, worst);
        failures += bad;
        p_mxv_destroy(h);
        free(st); free(dst); free(rew); free(drew); free(el); free(del); free(ep); free(dep); free(obs); free(dobs); free(fin); free(dfin);
        free(af); free(ai); free(te); free(tr); free(dte); free(dtr); free(fm);
    }
    /* Blackjack-v1 (mxv_bj_*): integer work — observations, rewards, masks, final observations bit for bit.  Actions: the oracle samples
     * them from the engine's action stream (actions = NULL) and the device gets what the oracle took. */
    {
        orc_bj_reset_t orc_bj_reset = (orc_bj_reset_t)must(orc, "orc_bj_reset");
        orc_bj_step_t orc_bj_step = (orc_bj_step_t)must(orc, "orc_bj_step");
        const int64_t nb = n;
        mxv_bj_config bc;
        memset(&bc, 0, sizeof bc);
        bc.device = 0; bc.natural = 1; bc.sab = 0; bc.max_episode_steps = 0; bc.num_envs = nb; bc.env_offset = (int64_t)env0;
        bc.seed = seed; bc.action_seed = action_seed;
        mxv_bj *hb = NULL;
        if (p_mxv_bj_create(&bc, &hb) != MXV_OK) {
            fprintf(stderr, "Blackjack-v1: mxv_bj_create: %s\n", p_mxv_bj_last_error(NULL));
            return 1;
        }
        int32_t *dealer = calloc(nb * 33, 4), *player = calloc(nb * 33, 4), *bel = calloc(nb, 4);
        int64_t *obs = malloc(8 * 3 * nb), *dobs = malloc(8 * 3 * nb), *fin = calloc(3 * nb, 8), *dfin = calloc(3 * nb, 8), *act = malloc(8 * nb);
        double *rew = malloc(8 * nb), *drew = malloc(8 * nb);
        uint8_t *te = malloc(nb), *tr = malloc(nb), *dte = malloc(nb), *dtr = malloc(nb), *fm = malloc(nb);
        orc_bj_reset(nb, env0, NULL, seed, 0, 1, NULL, dealer, player, bel, obs); /* the first reset call since seeding: ordinal 1 */
        int bad = p_mxv_bj_reset_host(hb, NULL, dobs) != MXV_OK || memcmp(obs, dobs, 8 * 3 * nb) != 0;
        long dones = 0;
        for (int t = 0; t < steps && !bad; ++t) {
            orc_bj_step(nb, env0, NULL, seed, action_seed, (uint64_t)t, 1, 0, 0, NULL, NULL, MXV_BJ_MAX_DRAWS, dealer, player, bel, act, obs, rew, te,
                        tr, fin, fm);
            if (p_mxv_bj_step_host(hb, act, NULL, dobs, drew, dte, dtr, dfin) != MXV_OK) {
                fprintf(stderr, "Blackjack-v1: mxv_bj_step_host: %s\n", p_mxv_bj_last_error(hb));
                return 1;
            }
            bad = memcmp(obs, dobs, 8 * 3 * nb) || memcmp(rew, drew, 8 * nb) || memcmp(te, dte, nb) || memcmp(tr, dtr, nb);
            for (int64_t i = 0; i < nb && !bad; ++i) {
                dones += fm[i];
                if (fm[i] && (fin[i] != dfin[i] || fin[nb + i] != dfin[nb + i] || fin[2 * nb + i] != dfin[2 * nb + i])) bad = 1;
            }
            if (bad) fprintf(stderr, "Blackjack-v1: mismatch at step %d\n", t);
        }
        uint64_t tt = 0;
        uint32_t rr = 0;
        p_mxv_bj_get_counters(hb, &tt, &rr);
        if (!bad && tt != (uint64_t)steps) bad = 1;
        printf("%-26s %s  envs=%lld steps=%d episodes_ended=%ld (every output bit for bit)\n", "Blackjack-v1", bad ? "FAILED" : "ok", (long long)nb,
               steps, dones);
        failures += bad;
        p_mxv_bj_destroy(hb);
        free(dealer); free(player); free(bel); free(obs); free(dobs); free(fin); free(dfin); free(act); free(rew); free(drew);
        free(te); free(tr); free(dte); free(dtr); free(fm);
    }
    printf(failures ? "abi_consumer: %d env kind(s) FAILED\n" : "abi_consumer: all five env kinds and Blackjack agree with the oracle\n", failures);
    return failures ? 1 : 0;
}
