"""P3 of SURVEY.md §8c: the device's random streams have the DISTRIBUTIONS of the reference's draws, at BASELINE.json's size.

Seeded streams cannot equal the reference's numbers (PCG64 there, Philox4x32-10 here: DESIGN.md §2), so what is mirrored is
the distribution of every draw on the path:
    reset states      np_random.uniform(low, high, size=(4,))         gym/envs/classic_control/cartpole.py:202 (and :154, :188, :160)
    Discrete actions  MultiDiscrete.sample = floor(random * nvec)      gym/spaces/multi_discrete.py:123
    Box actions       np_random.uniform(low, high).astype(float32)     gym/spaces/box.py:216-222
Tests: Kolmogorov-Smirnov against the uniform law per state component, chi-square on action frequencies, independence across
components / envs / consecutive steps (one Philox call serves 4 envs — and, for Discrete(2), 32 steps of each), and the episode
length distribution of a random CartPole policy against the reference's own (sampled here from the golden P2 run).
All thresholds are p > 1e-6 on fixed seeds: deterministic, and far from both tails for a healthy generator."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1 << 20


def _ks_uniform(x, lo, hi):
    from scipy import stats

    return stats.kstest((np.asarray(x, np.float64) - lo) / (hi - lo), "uniform").pvalue


def test_reset_states_are_uniform_and_independent_at_2_pow_20():
    from gym_amd import _native

    for kind, lo, hi, comps in ((_native.CARTPOLE, -0.05, 0.05, 4), (_native.ACROBOT, -0.1, 0.1, 4), (_native.MOUNTAINCAR, -0.6, -0.4, 1)):
        h = _native.Handle(kind, N, 500, seed=20260923, env_offset=1 << 22)
        h.reset_host()
        first, _ = h.get_state()
        h.reset_host()                                  # second draw of every env's own stream (reset ordinal 1)
        second, _ = h.get_state()
        for st in (first, second):
            x = st[:comps]
            assert x.min() >= lo and x.max() <= hi
            for c in range(comps):
                assert _ks_uniform(x[c], lo, hi) > 1e-6, (kind, c)
            # components of one env come from ONE Philox call, neighbouring envs from consecutive keys: no linear dependence
            if comps > 1:
                cc = np.corrcoef(x)
                assert np.abs(cc - np.eye(comps)).max() < 5.0 / np.sqrt(N), kind
            assert abs(np.corrcoef(x[0][:-1], x[0][1:])[0, 1]) < 5.0 / np.sqrt(N)
        assert abs(np.corrcoef(first[0], second[0])[0, 1]) < 5.0 / np.sqrt(N)      # successive resets of the same env
        assert not np.array_equal(first, second)
        h.close()
    p = _native.Handle(_native.PENDULUM, N, 200, seed=7)
    p.reset_host()
    st, _ = p.get_state()
    assert _ks_uniform(st[0], -np.pi, np.pi) > 1e-6 and _ks_uniform(st[1], -1.0, 1.0) > 1e-6
    p.close()


def test_action_frequencies_and_independence_at_2_pow_20():
    from scipy import stats

    from gym_amd.rollout import DeviceRollout

    K = 96                                              # three 32-step bit blocks for Discrete(2)
    r = DeviceRollout("CartPole-v1", N, seed=1, action_seed=2, env_offset=1 << 21)
    r.reset(seed=1)
    out = r.rollout_per_step(K)
    r.synchronize()                                     # the outputs are ready on the ENGINE's stream
    a = out["actions"].cpu().numpy().astype(np.int8)
    r.close()
    tot = a.size
    assert set(np.unique(a)) == {0, 1}
    assert abs(a.mean() - 0.5) < 5 * 0.5 / np.sqrt(tot)                              # global balance
    assert np.abs(a.mean(axis=1) - 0.5).max() < 6 * 0.5 / np.sqrt(N)                 # every step is balanced across envs
    per_env = a.sum(axis=0)                                                          # Binomial(96, 1/2) per env over time
    seen = np.bincount(per_env, minlength=K + 1)[24:73].astype(np.float64)           # the bins with an expected count >> 5
    law = stats.binom.pmf(np.arange(24, 73), K, 0.5)
    chi = stats.chisquare(seen, law / law.sum() * seen.sum())
    assert chi.pvalue > 1e-6
    x = a.astype(np.float32) - 0.5
    for lag in (1, 2, 31, 32, 33):                                                   # consecutive bits of a word, and across block borders
        assert abs((x[:-lag] * x[lag:]).mean()) * 4 < 5.0 / np.sqrt(x[:-lag].size), lag
    for d in (1, 2, 3, 4, 5):                                                        # the four envs of one Philox call, and the next group
        assert abs((x[:, :-d] * x[:, d:]).mean()) * 4 < 5.0 / np.sqrt(x[:, :-d].size), d

    for env_id, n_act in (("Acrobot-v1", 3), ("MountainCar-v0", 3)):
        r = DeviceRollout(env_id, N, seed=3, action_seed=4)
        r.reset(seed=3)
        out = r.rollout_per_step(24)
        r.synchronize()
        a = out["actions"].cpu().numpy()
        r.close()
        counts = np.bincount(a.ravel(), minlength=n_act)
        assert counts.size == n_act and stats.chisquare(counts).pvalue > 1e-6, env_id
        pair = np.bincount((a[:-1] * n_act + a[1:]).ravel(), minlength=n_act * n_act)   # successive actions of one env
        assert stats.chisquare(pair).pvalue > 1e-6, env_id
        quad = np.bincount((a[:, 0::4] * n_act + a[:, 1::4]).ravel(), minlength=n_act * n_act)  # neighbours inside one call
        assert stats.chisquare(quad).pvalue > 1e-6, env_id

    for env_id, lim in (("Pendulum-v1", 2.0), ("MountainCarContinuous-v0", 1.0)):
        r = DeviceRollout(env_id, N, seed=5, action_seed=6)
        r.reset(seed=5)
        out = r.rollout_per_step(8)
        r.synchronize()
        a = out["actions"].cpu().numpy().astype(np.float64)
        r.close()
        assert a.min() >= -lim and a.max() <= lim
        assert _ks_uniform(a.ravel()[: 1 << 22], -lim, lim) > 1e-6, env_id
        assert abs(np.corrcoef(a[0], a[1])[0, 1]) < 5.0 / np.sqrt(N)


def test_random_policy_cartpole_episode_lengths_follow_the_references_distribution():
    """Episode lengths of a uniformly random policy are a property of dynamics x action law x reset law together.  Reference
    sample: the CartPole P2 golden (gym's own SyncVectorEnv under action_space.sample(), 450 steps x 8 envs: ~160 episodes);
    device sample: 2^20 envs x 128 steps (millions of episodes).  Two-sample Kolmogorov-Smirnov."""
    from scipy import stats

    from helpers import load_golden
    from gym_amd.rollout import DeviceRollout

    g = load_golden("CartPole", "p2_default")
    ref_len = g["ep_length"][g["ep_mask"].astype(bool)]
    assert ref_len.size > 100
    r = DeviceRollout("CartPole-v1", N, seed=8, action_seed=9)
    r.enable_episode_stats()
    r.reset(seed=8)
    out = r.rollout_per_step(128)
    r.synchronize()
    done = (out["terminated"] | out["truncated"]).cpu().numpy().astype(bool)
    lens = out["ep_length"].cpu().numpy()[done]
    r.close()
    # drop the censoring of the 128-step window: keep episodes that STARTED in the first 64 steps (lengths > 64 are < 1e-3 of all)
    steps = np.nonzero(done)[0]
    started = steps - lens + 1
    sample = lens[(started >= 0) & (started < 64)]
    assert sample.size > 1_000_000
    assert abs(sample.mean() - ref_len.mean()) < 4 * ref_len.std() / np.sqrt(ref_len.size)
    assert stats.ks_2samp(ref_len, sample[:: max(1, sample.size // 200_000)]).pvalue > 1e-4
