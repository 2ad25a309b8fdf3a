"""The HOST side of the toy_text adapters on the GPU-less box: gym_amd._native.Tab / Blackjack replaced by the oracle-backed FakeTab / FakeBlackjack
(tests/oracle_engine.py), then the very test bodies the GPU suite runs on the device (tests/test_gpu_toytext.py,
test_gpu_toytext_stats.py, test_gpu_toytext_normalize.py): spaces, infos and their dtype quirks, errors, Taxi's helpers through call(),
pickling, the statistics and normalisation wrappers against the reference's own runs (goldens)."""
import numpy as np
import pytest


@pytest.fixture
def fake_tab(monkeypatch):
    from gym_amd import _native
    from oracle_engine import FakeBlackjack, FakeTab

    monkeypatch.setattr(_native, "Tab", FakeTab)
    monkeypatch.setattr(_native, "Blackjack", FakeBlackjack)


def test_tabular_vector_env_contract_on_the_oracle_backed_handle(fake_tab):
    import test_gpu_toytext as t

    t.test_hip_tabular_vector_env_contract()
    t.test_taxi_helpers_through_call()
    for gid in ("FrozenLake-v1", "Taxi-v3"):
        t.test_pickle_round_trip_continues_identically(gid)


def test_blackjack_vector_env_contract_on_the_oracle_backed_handle(fake_tab):
    import test_gpu_blackjack as t

    t.test_hip_blackjack_vector_env_contract()
    t.test_pickle_round_trip_continues_identically()


@pytest.mark.parametrize("tag", ["FrozenLake-v1", "Taxi-v3", "Blackjack-v1"])
def test_statistics_wrappers_over_the_tabular_adapters_replay_the_reference(fake_tab, tag):
    import test_gpu_toytext_stats as t

    t.test_wrappers_over_the_toy_text_adapters_report_what_the_reference_reports(tag)


@pytest.mark.parametrize("tag", ["FrozenLake-v1", "Taxi-v3", "Blackjack-v1"])
def test_per_sub_env_normalize_reward_over_the_tabular_adapters_bit_for_bit(fake_tab, tag, monkeypatch):
    import test_gpu_toytext_normalize as t

    t.test_per_sub_env_normalize_reward_over_the_toy_text_engines_bit_for_bit(tag, False, monkeypatch)


def _single_env_walk(make_single, make_vector, gid, kw, steps=400):
    """HipToyTextEnv against a one-env VECTOR env of the same engine with the same seed and actions: the single env hands back the terminal
    observation / info where the vector env reports final_observation / final_info, and its next reset() returns what the vector env
    returned as the (auto)reset observation — the engine drew it once, nobody resets twice."""
    one, vec = make_single(gid, **kw), make_vector(gid, 1, **kw)
    o, i = one.reset(seed=9)
    vo, vi = vec.reset(seed=9)
    first = (lambda x: (int(x[0][0]), int(x[1][0]), bool(x[2][0]))) if gid == "Blackjack-v1" else (lambda x: int(x[0]))
    assert o == first(vo) and (i == {} if gid == "Blackjack-v1" else i["prob"] == vi["prob"][0])
    one.action_space.seed(4)
    episodes = 0
    for _ in range(steps):
        a = one.action_space.sample()
        o, r, te, tr, info = one.step(a)
        vo, vr, vte, vtr, vinfo = vec.step(np.array([a]))
        assert (r, te, tr) == (float(vr[0]), bool(vte[0]), bool(vtr[0])) and isinstance(r, float) and isinstance(te, bool)
        if te or tr:
            fin = vinfo["final_observation"][0]
            want = vinfo["final_info"][0] or {}
            assert o == (fin if isinstance(fin, tuple) else int(fin)) and set(info) == set(want) and all(np.array_equal(info[k], want[k]) for k in want)
            o2, i2 = one.reset()
            assert o2 == first(vo) and isinstance(i2, dict)
            episodes += 1
        else:
            assert o == first(vo)
            if gid != "Blackjack-v1":
                assert info["prob"] == vinfo["prob"][0]
    assert episodes > 3
    assert one.reset(seed=9)[0] == first(vec.reset(seed=9)[0])          # a seeded reset is a real one
    one.close(), vec.close()
    return episodes


CASES = [("FrozenLake-v1", {}), ("FrozenLake8x8-v1", {"max_episode_steps": 30}), ("Taxi-v3", {"max_episode_steps": 25}),
         ("CliffWalking-v0", {"max_episode_steps": 20}), ("Blackjack-v1", {"natural": True, "sab": False})]


@pytest.mark.parametrize("gid,kw", CASES)
def test_toy_text_single_env_equals_a_one_env_vector_env(fake_tab, gid, kw):
    import gym_amd
    from gym_amd.single_env import HipToyTextEnv

    _single_env_walk(HipToyTextEnv, gym_amd.make, gid, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("gid,kw", CASES)
def test_toy_text_single_env_on_the_device(gid, kw):
    import pickle

    import gym_amd
    from gym_amd.single_env import HipToyTextEnv

    _single_env_walk(HipToyTextEnv, gym_amd.make, gid, kw)
    env = HipToyTextEnv(gid, **kw)
    env.reset(seed=2)
    env.action_space.seed(1)
    for _ in range(7):
        if any(env.step(env.action_space.sample())[2:4]):
            env.reset()
    twin = pickle.loads(pickle.dumps(env))
    (o1, i1), (o2, i2) = env.reset(), twin.reset()
    assert o1 == o2 and set(i1) == set(i2) and all(np.array_equal(i1[k], i2[k]) for k in i1)
    a = env.action_space.sample()
    x, y = env.step(a), twin.step(a)
    assert x[:4] == y[:4]
    env.close(), twin.close()
