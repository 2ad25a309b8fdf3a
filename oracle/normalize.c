/*
 * oracle/normalize.c — CPU restatement of gym.wrappers.NormalizeObservation / NormalizeReward
 *  for a vector env.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and by nothing under
 * gym_amd/); it is the checker of the HIP normalisation kernels, never a fallback for them.
 *
 * Follows /root/reference/gym/wrappers/normalize.py:
 *   RunningMeanStd.update                     :17-22   batch_mean = np.mean(x, axis=0); batch_var = np.var(x, axis=0)
 *   update_mean_var_count_from_moments        :32-47
 *   NormalizeObservation.normalize            :90-93   (obs - mean) / np.sqrt(var + epsilon)      -> float64
 *   NormalizeReward.step / .normalize         :127-145 returns = returns*gamma + rews; update(returns);
 *                                                      rews / np.sqrt(var + epsilon); returns[dones] = 0
 * and NumPy 2.2.6's arithmetic for those calls (numpy/_core/_methods.py _mean/_var; pinned bit-exact against
 * tests/golden/normalize_*.npz, which tests/golden/make_golden_normalize.py generates from the live reference):
 *   - np.mean / np.var of a C-contiguous float32 (N,O) array over axis 0 accumulate ROW BY ROW IN FLOAT32
 *     (the reduction axis is the outer loop of the nditer), then divide as double(sum32)/double(N) cast back to float32
 *     (true_divide with an intp count and out=float32, casting='unsafe');
 *   - np.mean / np.var of a 1-D float64 array use NumPy's pairwise summation (blocks of 128, 8 accumulators);
 *   - batch_var(float32 array) * batch_count(python int) stays float32 (NEP 50 weak scalar), everything else of
 *     update_mean_var_count_from_moments is float64 in source order.
 *
 * mode 0 = that arithmetic (what the goldens pin).  mode 1 = the DEVICE's definition of the batch moments: exact sums
 * (long double here, fp64 trees on the GPU) of x and x*x, rounded to the reference's float32 moment dtype for
 * observations — i.e. the same quantities without the float32 accumulation error the reference itself carries.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* numpy/_core/src/umath/loops_utils.h.src: @TYPE@_pairwise_sum, contiguous double input */
static double pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* update_mean_var_count_from_moments, normalize.py:32-47, one column; m_b is passed in already formed
 * (float32 product for observation statistics, float64 product for the scalar return statistics). */
static void merge_moments(double *mean, double *var, double count, double batch_mean, double m_b, double batch_count) {
    const double delta = batch_mean - *mean;                                /* :36 */
    const double tot_count = count + batch_count;                           /* :37 */
    const double new_mean = *mean + delta * batch_count / tot_count;        /* :39 */
    const double m_a = *var * count;                                        /* :40 */
    const double M2 = m_a + m_b + delta * delta * count * batch_count / tot_count; /* :42 */
    *mean = new_mean;
    *var = M2 / tot_count;                                                  /* :43 */
}

/* ---- the device's definition, split the way a sharded vector env needs it (include/mxv.h mxv_norm_*_sums / *_apply) ---- */

/* per-step column sums of one shard: sums[K][2*O] = (sum_0..sum_{O-1}, sumsq_0..sumsq_{O-1}), accumulated in long double
 * (x87 80-bit) and rounded once to double — the quantity the GPU forms with its fp64 trees. */
void orc_norm_obs_sums(const float *x, int64_t K, int64_t n, int O, double *sums) {
    for (int64_t k = 0; k < K; ++k)
        for (int j = 0; j < O; ++j) {
            long double S = 0, Q = 0;
            for (int64_t i = 0; i < n; ++i) {
                const long double v = x[(k * n + i) * O + j];
                S += v;
                Q += v * v;
            }
            sums[k * 2 * O + j] = (double)S;
            sums[k * 2 * O + O + j] = (double)Q;
        }
}

static void tree_over_ranks(const double *all_sums, int W, int64_t K, int V, int64_t k, int idx, double *out) {
    double b[64];
    for (int w = 0; w < W; ++w) b[w] = all_sums[((int64_t)w * K + k) * V + idx];
    for (int stride = 1; stride < W; stride <<= 1)
        for (int w = 0; w + stride < W; w += 2 * stride) b[w] += b[w + stride];
    *out = b[0];
}

/* batch moments in the reference's float32 moment dtype from the (merged) exact sums, then the running update */
static void obs_update_from_sums(double *mean, double *var, double count, double S, double Q, double N) {
    const float mean32 = (float)(S / N);
    const double m = (double)mean32;
    double v = ((Q - 2.0 * m * S) + N * m * m) / N;
    if (!(v > 0.0)) v = 0.0;
    const float var32 = (float)v;
    merge_moments(mean, var, count, m, (double)(var32 * (float)N), N);
}

void orc_norm_obs_apply(double *mean, double *var, double *count, double epsilon, const float *x, int64_t K, int64_t n,
                        int O, const double *all_sums, int W, int64_t total_rows, double *y) {
    for (int64_t k = 0; k < K; ++k) {
        for (int j = 0; j < O; ++j) {
            double S, Q;
            tree_over_ranks(all_sums, W, K, 2 * O, k, j, &S);
            tree_over_ranks(all_sums, W, K, 2 * O, k, O + j, &Q);
            obs_update_from_sums(&mean[j], &var[j], *count, S, Q, (double)total_rows);
        }
        *count += (double)total_rows;
        for (int j = 0; j < O; ++j) {
            const double denom = sqrt(var[j] + epsilon);
            for (int64_t i = 0; i < n; ++i) y[(k * n + i) * O + j] = ((double)x[(k * n + i) * O + j] - mean[j]) / denom;
        }
    }
}

void orc_norm_reward_sums(double *returns, const double *rew, const uint8_t *term, const uint8_t *trunc, int64_t K,
                          int64_t n, double gamma, double *sums) {
    for (int64_t k = 0; k < K; ++k) {
        long double S = 0, Q = 0;
        for (int64_t i = 0; i < n; ++i) {
            returns[i] = returns[i] * gamma + rew[k * n + i]; /* :132 */
            S += returns[i];
            Q += (long double)returns[i] * returns[i];
        }
        sums[k * 2] = (double)S;
        sums[k * 2 + 1] = (double)Q;
        for (int64_t i = 0; i < n; ++i)
            if (term[k * n + i] | trunc[k * n + i]) returns[i] = 0.0; /* :135-136 */
    }
}

void orc_norm_reward_apply(double *mean, double *var, double *count, double epsilon, const double *rew, int64_t K, int64_t n,
                           const double *all_sums, int W, int64_t total_rows, double *out) {
    const double N = (double)total_rows;
    for (int64_t k = 0; k < K; ++k) {
        double S, Q;
        tree_over_ranks(all_sums, W, K, 2, k, 0, &S);
        tree_over_ranks(all_sums, W, K, 2, k, 1, &Q);
        const double bmean = S / N;
        double v = Q / N - bmean * bmean;
        if (!(v > 0.0)) v = 0.0;
        merge_moments(mean, var, *count, bmean, v * N, N);
        *count += N;
        const double denom = sqrt(*var + epsilon);
        for (int64_t i = 0; i < n; ++i) out[k * n + i] = rew[k * n + i] / denom;
    }
}

/* ---- the reference's arithmetic (mode 0) and the one-call forms ------------------------------------------------------- */

/* RunningMeanStd(shape=(O,)).update(x) for x float32 [n][O] exactly as NumPy evaluates it; count is shared by the columns. */
static void rms_update_obs_reference(double *mean, double *var, double *count, const float *x, int64_t n, int O) {
    for (int j = 0; j < O; ++j) {
        float acc = 0.0f;
        for (int64_t i = 0; i < n; ++i) acc = acc + x[i * O + j];
        const float bmean = (float)((double)acc / (double)n);
        float acc2 = 0.0f;
        for (int64_t i = 0; i < n; ++i) {
            float d = x[i * O + j] - bmean;
            d = d * d;
            acc2 = acc2 + d;
        }
        const float bvar = (float)((double)acc2 / (double)n);
        const double m_b = (double)(bvar * (float)n); /* float32 array * python int -> float32 (:41) */
        merge_moments(&mean[j], &var[j], *count, (double)bmean, m_b, (double)n);
    }
    *count += (double)n;
}

/* K consecutive NormalizeObservation.normalize calls: x float32 [K][n][O] -> y float64 [K][n][O]. */
void orc_norm_obs_batches(double *mean, double *var, double *count, double epsilon, const float *x, int64_t K, int64_t n,
                          int O, int mode, double *y) {
    if (mode != 0) {
        double *sums = (double *)malloc(sizeof(double) * (size_t)(K * 2 * O));
        orc_norm_obs_sums(x, K, n, O, sums);
        orc_norm_obs_apply(mean, var, count, epsilon, x, K, n, O, sums, 1, n, y);
        free(sums);
        return;
    }
    for (int64_t k = 0; k < K; ++k) {
        const float *xk = x + k * n * O;
        rms_update_obs_reference(mean, var, count, xk, n, O);
        for (int j = 0; j < O; ++j) {
            const double denom = sqrt(var[j] + epsilon);
            for (int64_t i = 0; i < n; ++i) y[(k * n + i) * O + j] = ((double)xk[i * O + j] - mean[j]) / denom;
        }
    }
}

/* K consecutive NormalizeReward.step calls (:127-145): rew float64 [K][n], flags uint8 [K][n] -> out float64 [K][n].
 * returns[n] is the wrapper's discounted-return accumulator (zeros at construction). */
void orc_norm_reward_steps(double *returns, double *mean, double *var, double *count, double gamma, double epsilon,
                           const double *rew, const uint8_t *term, const uint8_t *trunc, int64_t K, int64_t n, int mode,
                           double *out) {
    if (mode != 0) {
        double *sums = (double *)malloc(sizeof(double) * (size_t)(K * 2));
        orc_norm_reward_sums(returns, rew, term, trunc, K, n, gamma, sums);
        orc_norm_reward_apply(mean, var, count, epsilon, rew, K, n, sums, 1, n, out);
        free(sums);
        return;
    }
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t k = 0; k < K; ++k) {
        const double *r = rew + k * n;
        for (int64_t i = 0; i < n; ++i) returns[i] = returns[i] * gamma + r[i]; /* :132 */
        const double bmean = pairwise_sum(returns, n) / (double)n;
        for (int64_t i = 0; i < n; ++i) {
            const double d = returns[i] - bmean;
            tmp[i] = d * d;
        }
        const double bvar = pairwise_sum(tmp, n) / (double)n;
        merge_moments(mean, var, *count, bmean, bvar * (double)n, (double)n);
        *count += (double)n;
        const double denom = sqrt(*var + epsilon); /* :144-145 */
        for (int64_t i = 0; i < n; ++i) out[k * n + i] = r[i] / denom;
        for (int64_t i = 0; i < n; ++i)
            if (term[k * n + i] | trunc[k * n + i]) returns[i] = 0.0; /* :135-136 */
    }
    free(tmp);
}

/* ---- the PER-SUB-ENV wrappers: gym.vector.make(id, n, wrappers=[NormalizeObservation, NormalizeReward]) -----------------------
 * gym/vector/__init__.py:56-65 wraps EVERY sub-env, so each sub-env owns a RunningMeanStd that RunningMeanStd.update (:17-22) feeds
 * with a batch of ONE row: np.array([obs]) (:78, :88): batch_mean = np.mean(x, axis=0) = the row itself (float32 for observations:
 * the sum of one float32 over a count of 1), batch_var = np.var(x, axis=0) = float32 0, batch_count = 1.
 * update_mean_var_count_from_moments (:32-47) then runs in float64 in source order; m_b = float32(0) * 1 adds an exact zero.
 * Checked bit for bit against the reference's own run: tests/golden/vector_make_normalize_*.npz through
 * tests/test_normalize_oracle.py::test_subnorm_oracle_replays_the_reference. */
static void sub_update(double *mean, double *var, double *count, double x) {
    const double delta = x - *mean;                                         /* :36 */
    const double tot = *count + 1.0;                                        /* :37 */
    const double new_mean = *mean + delta * 1.0 / tot;                      /* :39 */
    const double m_a = *var * *count;                                       /* :40 */
    const double m_b = (double)(0.0f * 1.0f);                               /* :41 */
    const double M2 = m_a + m_b + delta * delta * *count * 1.0 / tot;       /* :42 */
    *mean = new_mean;
    *var = M2 / tot;                                                        /* :43 */
    *count = tot;                                                           /* :44 */
}

/* NormalizeObservation around every sub-env under SyncVectorEnv's autoreset (sync_vector_env.py:142-156): where a sub-env's
 * episode ended, its step() normalised the TERMINAL observation (fin; -> yfin, float64, info["final_observation"]), then its
 * reset() normalised the reset observation (the row of the batch).  x, fin: float32 [K][n][O]; te, tr: uint8 [K][n] or NULL
 * (reset()); mean, var: [n][O]; count: [n]; y: float64 [K][n][O]; yfin: float64 [K][n][O] (rows of finished sub-envs) or NULL. */
void orc_subnorm_obs(double *mean, double *var, double *count, double epsilon, const float *x, const float *fin,
                     const uint8_t *te, const uint8_t *tr, int64_t K, int64_t n, int O, double *y, double *yfin) {
    for (int64_t k = 0; k < K; k++)
        for (int64_t i = 0; i < n; i++) {
            const int64_t r = k * n + i;
            const int done = te && (te[r] | tr[r]);
            if (done && fin) {
                double c = count[i];
                for (int j = 0; j < O; j++) {
                    c = count[i];
                    sub_update(&mean[i * O + j], &var[i * O + j], &c, (double)fin[r * O + j]);
                }
                count[i] = c;
                if (yfin)
                    for (int j = 0; j < O; j++)
                        yfin[r * O + j] = ((double)fin[r * O + j] - mean[i * O + j]) / sqrt(var[i * O + j] + epsilon); /* :93 */
            }
            double c = count[i];
            for (int j = 0; j < O; j++) {
                c = count[i];
                sub_update(&mean[i * O + j], &var[i * O + j], &c, (double)x[r * O + j]);
            }
            count[i] = c;
            for (int j = 0; j < O; j++) y[r * O + j] = ((double)x[r * O + j] - mean[i * O + j]) / sqrt(var[i * O + j] + epsilon);
        }
}

/* NormalizeReward around every sub-env (:127-145): returns = returns * gamma + rews; return_rms.update(returns) with a batch of one;
 * rews / sqrt(var + epsilon); returns = 0 where the episode ended.  All [n] / [K][n] float64. */
void orc_subnorm_rew(double *returns, double *mean, double *var, double *count, double gamma, double epsilon, const double *rew,
                     const uint8_t *te, const uint8_t *tr, int64_t K, int64_t n, double *out) {
    for (int64_t k = 0; k < K; k++)
        for (int64_t i = 0; i < n; i++) {
            const int64_t r = k * n + i;
            returns[i] = returns[i] * gamma + rew[r];                       /* :132 */
            sub_update(&mean[i], &var[i], &count[i], returns[i]);           /* :144 */
            out[r] = rew[r] / sqrt(var[i] + epsilon);                       /* :145 */
            if (te[r] | tr[r]) returns[i] = 0.0;                            /* :134-135 */
        }
}
