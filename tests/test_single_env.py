"""gym_amd.single_env.HipEnv — gym.Env's single-environment contract (gym/core.py:75-184) over the engine, what `gym.make("hip/<id>")`
returns when no num_envs is given.  CPU: over the oracle-backed handle (tests/oracle_engine.FakeHandle), against the
LIVE reference where it is importable.  GPU: the same checks on the device (`-m gpu`)."""
import numpy as np
import pytest

IDS = ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0"]


def _readme_loop(env, steps, seed=42):
    """/root/reference/README.md:29-41 with a random policy; returns what happened."""
    env.action_space.seed(seed)
    observation, info = env.reset(seed=seed)
    log, episodes = [], 0
    for _ in range(steps):
        action = env.action_space.sample()
        observation, reward, terminated, truncated, info = env.step(action)
        log.append((observation.copy(), reward, terminated, truncated))
        assert isinstance(observation, np.ndarray) and observation.dtype == np.float32 and observation.shape == env.observation_space.shape
        assert isinstance(reward, float) and isinstance(terminated, bool) and isinstance(truncated, bool) and info == {}
        if terminated or truncated:
            observation, info = env.reset()
            episodes += 1
    return log, episodes


def _checks(make_env, have_reference):
    from gym_amd import error

    # (1) the README loop, and determinism as tests/envs/test_envs.py:63-115 asks it: same seed -> same rollout, another seed -> another
    for gid in IDS:
        a, b, c = make_env(gid), make_env(gid), make_env(gid)
        la, ea = _readme_loop(a, 600)
        lb, eb = _readme_loop(b, 600)
        lc, _ = _readme_loop(c, 600, seed=43)
        assert ea == eb and (ea > 0 or gid in ("MountainCarContinuous-v0",)), (gid, ea)
        for x, y in zip(la, lb):
            assert np.array_equal(x[0], y[0]) and x[1:] == y[1:]
        assert any(not np.array_equal(x[0], z[0]) for x, z in zip(la, lc))
        a.close(), b.close(), c.close()
    # (2) TimeLimit, OrderEnforcing, the action assert, CartPole's steps_beyond_terminated
    env = make_env("CartPole-v1", max_episode_steps=5)
    with pytest.raises(error.ResetNeeded):
        env.step(0)
    env.reset(seed=1)
    flags = [env.step(t % 2)[2:4] for t in range(5)]
    assert [f[1] for f in flags] == [False, False, False, False, True] and not any(f[0] for f in flags)     # truncated at exactly 5 steps
    assert env.step(0)[3] is True                                  # ... and on every later step until reset() (time_limit.py:50-54)
    obs, _ = env.reset()
    assert env.step(1)[3] is False and np.abs(obs).max() <= 0.05
    for bad in (2, -1, 0.5, np.array([0, 1])):
        with pytest.raises(AssertionError):
            env.step(bad)
    env.close()
    with pytest.raises(error.ClosedEnvironmentError):
        env.step(0)
    env = make_env("CartPole-v1")
    env.reset(seed=3)
    rewards, term = [], False
    while not term:
        _, r, term, _, _ = env.step(1)                             # push right until the pole falls
        rewards.append(r)
    assert rewards[-1] == 1.0 and env.step(1)[1] == 0.0 and env.step(1)[2] is True       # cartpole.py:169-184
    assert env.gravity == 9.8 and env.state.shape == (4,) and env.spec.id.endswith("CartPole-v1")
    env.close()
    # (3) Box actions: shape (1,) array-likes, out-of-range values clipped by the env itself
    env = make_env("Pendulum-v1")
    env.reset(seed=0)
    o1 = env.step(np.array([5.0], dtype=np.float32))
    env.reset(seed=0)
    o2 = env.step(np.array([2.0], dtype=np.float32))
    assert np.array_equal(o1[0], o2[0])
    env.close()
    if not have_reference:
        return
    # (4) against the reference's own single envs: its state injected, the same actions, step by step (dynamics, flags at the limit,
    #     rewards after termination) — the engine's goldens hold the numerics; this holds the single-env PLUMBING
    import gym

    for gid in IDS:
        ref, mine = gym.make(gid), make_env(gid)
        ref.action_space.seed(5)
        ref.reset(seed=5)
        mine.reset(seed=5)
        st = np.asarray(ref.unwrapped.state, dtype=np.float64)
        mine._vec.handle.set_state(st.reshape(-1, 1), np.zeros(1, np.int32))
        for t in range(260):
            act = ref.action_space.sample()
            ro, rr, rte, rtr, _ = ref.step(act)
            mo, mr, mte, mtr, _ = mine.step(act)
            assert (rte, rtr) == (mte, mtr), (gid, t)
            np.testing.assert_allclose(mo, ro, rtol=1e-5, atol=1e-7)
            assert mr == pytest.approx(float(rr), rel=1e-12, abs=1e-9)
            if rte or rtr:
                ref.reset()
                mine.reset()
                st = np.asarray(ref.unwrapped.state, dtype=np.float64)
                mine._vec.handle.set_state(st.reshape(-1, 1), np.zeros(1, np.int32))
        ref.close(), mine.close()


def test_single_env_contract_over_the_oracle_backed_handle(monkeypatch):
    from gym_amd import _native
    from gym_amd.single_env import HipEnv
    from oracle_engine import FakeHandle

    monkeypatch.setattr(_native, "Handle", FakeHandle)
    try:
        from test_host_logic import _ref_gym

        _ref_gym()
        have = True
    except BaseException:      # pytest.skip inside _ref_gym where the reference is not importable
        have = False
    _checks(lambda gid, **kw: HipEnv(gid, **kw), have)


def test_gym_make_without_num_envs_is_a_gym_env(monkeypatch):
    """gym.make("hip/CartPole-v1") — no num_envs, as the reference's README writes it — returns a gym.Env (not a vector env) with
    gym.spaces spaces; the README loop runs unchanged; num_envs=... still gives the vector env; toy_text ids say what to do instead."""
    from test_host_logic import _ref_gym

    gym = _ref_gym()
    from gym_amd import _native, plugin
    from oracle_engine import FakeHandle, FakeTab

    monkeypatch.setattr(_native, "Handle", FakeHandle)
    monkeypatch.setattr(_native, "Tab", FakeTab)
    plugin.register_envs(gym)
    env = gym.make("hip/CartPole-v1")
    assert isinstance(env, gym.Env) and not isinstance(env, gym.vector.VectorEnv) and not getattr(env, "is_vector_env", False)
    assert isinstance(env.action_space, gym.spaces.Discrete) and isinstance(env.observation_space, gym.spaces.Box)
    assert env.observation_space == gym.make("CartPole-v1").observation_space and env.spec.id == "hip/CartPole-v1"
    log, episodes = _readme_loop(env, 300)
    assert episodes > 5
    env.close()
    assert isinstance(gym.make("hip/CartPole-v1", num_envs=3), gym.vector.VectorEnv)
    lake = gym.make("hip/FrozenLake-v1")          # the toy_text ids: HipToyTextEnv (tests/test_toytext_host_logic.py)
    assert isinstance(lake, gym.Env) and isinstance(lake.observation_space, gym.spaces.Discrete) and lake.spec.max_episode_steps == 100
    lake.close()
    # what gym.make itself wraps around an env works on top of the single env like on any gym.Env (gym/envs/registration.py:676-695)
    auto = gym.make("hip/CartPole-v1", autoreset=True, max_episode_steps=9)
    assert type(auto).__name__ == "AutoResetWrapper" and type(auto.env).__name__ == "TimeLimit"
    auto.reset(seed=3)
    ends = 0
    for _ in range(60):
        obs, r, te, tr, info = auto.step(auto.action_space.sample())
        if te or tr:
            ends += 1
            assert info["final_observation"].shape == (4,) and "final_info" in info
    assert ends >= 6 and obs.dtype == np.float32 and isinstance(r, float) and isinstance(te, bool)
    auto.close()
    short = gym.make("hip/Pendulum-v1", time_limit=7)
    short.reset(seed=0)
    assert [short.step(short.action_space.sample())[3] for _ in range(7)] == [False] * 6 + [True]
    short.close()


def test_the_references_env_checker_accepts_the_single_env(monkeypatch):
    """gym.utils.env_checker.check_env (the checker gym.make runs on every env it builds, gym/utils/env_checker.py:231-299) on the engine's
    single env, every classic-control id: spaces, reset(seed) determinism and the np_random contract, reset(options), step / reward / info
    types.  And gym.make's default stack (PassiveEnvChecker included) steps it without a complaint."""
    import warnings

    from test_host_logic import _ref_gym

    gym = _ref_gym()
    from gym.utils.env_checker import check_env

    from gym_amd import _native, plugin
    from oracle_engine import FakeHandle

    monkeypatch.setattr(_native, "Handle", FakeHandle)
    plugin.register_envs(gym)
    for gid in ("CartPole-v1", "Pendulum-v1", "MountainCarContinuous-v0", "Acrobot-v1", "MountainCar-v0"):
        env = gym.make("hip/" + gid, disable_env_checker=True)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            check_env(env.unwrapped, skip_render_check=True)
        msgs = [str(x.message) for x in w if "symmetric and normalized" not in str(x.message)]      # (Pendulum's [-2, 2]: the reference env's own)
        assert not msgs, (gid, msgs)
        a, b = env.reset(seed=5)[0], env.reset(seed=5)[0]
        assert np.array_equal(a, b) and env.unwrapped.np_random.integers(1 << 30) == type(env.unwrapped.np_random)(np.random.PCG64(np.random.SeedSequence(5))).integers(1 << 30)
        env.close()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env = gym.make("hip/CartPole-v1")            # PassiveEnvChecker on top, as gym.make builds it by default
        env.reset(seed=0)
        for _ in range(20):
            _, _, te, tr, _ = env.step(env.action_space.sample())
            if te or tr:
                env.reset()
        env.close()
    assert not [str(x.message) for x in w if "hip/" not in str(x.message) and "symmetric" not in str(x.message)], [str(x.message) for x in w]


@pytest.mark.gpu
def test_single_env_contract_on_the_device():
    from gym_amd.single_env import HipEnv

    _checks(lambda gid, **kw: HipEnv(gid, **kw), False)


@pytest.mark.gpu
@pytest.mark.parametrize("gid", ["CartPole-v1", "Pendulum-v1", "Acrobot-v1"])
def test_single_env_pickles_and_continues_like_the_original(gid):
    """The reference's tests/envs/test_envs.py::test_pickle_env on the device: a pickled copy's next (unseeded) reset and step equal the
    original's — the copy carries the engine's state, counters and Philox positions (Handle.snapshot)."""
    import pickle

    from gym_amd.single_env import HipEnv

    env = HipEnv(gid)
    env.reset(seed=11)
    env.action_space.seed(3)
    for _ in range(5):
        env.step(env.action_space.sample())
    twin = pickle.loads(pickle.dumps(env))
    (o1, i1), (o2, i2) = env.reset(), twin.reset()
    assert np.array_equal(o1, o2) and i1 == i2 == {}
    a = env.action_space.sample()
    s1, s2 = env.step(a), twin.step(a)
    assert np.array_equal(s1[0], s2[0]) and s1[1:] == s2[1:]
    env.close(), twin.close()
