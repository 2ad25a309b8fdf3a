"""Interop with the reference's own classes (gym_amd/interop.py), checked against the LIVE reference in the build container
(no device: the engine handle is replaced by an oracle-backed stand-in, tests/oracle_engine.FakeHandle).

What the reference's code checks by identity and what therefore must hold for an env obtained through gym.make("hip/<id>"):
    assert isinstance(env, VectorEnv)            gym/vector/vector_env.py:289
    isinstance(env.action_space, Box)            gym/wrappers/clip_action.py:28
    except gym.error.ResetNeeded                 user code
and the reference's own wrappers must run on top of the engine: gym.vector.VectorEnvWrapper (vector_env.py:277-332),
gym.wrappers.RecordEpisodeStatistics (record_episode_statistics.py:40-151), gym.wrappers.VectorListInfo."""
import pickle

import numpy as np
import pytest

from test_host_logic import _ref_gym


@pytest.fixture(params=["dense", "packed"])
def hip(monkeypatch, request):
    """The engine handle replaced by the oracle-backed stand-in; "packed" = the stand-in of a LARGE vector env (final
    observations arrive as packed records, infos are built lazily from them)."""
    gym = _ref_gym()
    from oracle_engine import FakeHandle, PackedFakeHandle

    from gym_amd import _native, plugin

    monkeypatch.setattr(_native, "Handle", PackedFakeHandle if request.param == "packed" else FakeHandle)
    plugin.register_envs(gym)
    return gym


def test_gym_make_returns_an_instance_of_the_references_vector_env(hip):
    gym = hip
    env = gym.make("hip/CartPole-v1", num_envs=6)
    from gym_amd.vector_env import HipVectorEnv

    assert isinstance(env, gym.vector.VectorEnv) and isinstance(env, gym.Env) and isinstance(env, HipVectorEnv)
    assert type(env).__name__ == "HipVectorEnv" and env.unwrapped is env and env.num_envs == 6 and env.is_vector_env
    assert env.spec.id == "hip/CartPole-v1"                       # gym.make stamps its own spec (registration.py:657)
    # spaces are the reference's classes, equal to what SyncVectorEnv builds for the same id
    ref = gym.vector.make("CartPole-v1", num_envs=6, asynchronous=False)
    for name in ("observation_space", "action_space", "single_observation_space", "single_action_space"):
        mine, theirs = getattr(env, name), getattr(ref, name)
        assert type(mine) is type(theirs) and mine == theirs, name
    assert isinstance(env.action_space, gym.spaces.MultiDiscrete) and isinstance(env.single_observation_space, gym.spaces.Box)
    env.action_space.seed(3), ref.action_space.seed(3)
    assert np.array_equal(env.action_space.sample(), ref.action_space.sample())
    ref.close()
    # the step contract through the reference-typed object
    obs, info = env.reset(seed=1)
    assert obs.shape == (6, 4) and obs.dtype == np.float32 and info == {}
    obs, rew, term, trunc, infos = env.step(env.action_space.sample())
    assert rew.dtype == np.float64 and term.dtype == np.bool_ and isinstance(infos, dict)
    env.close()


def test_engine_errors_are_the_references_errors(hip):
    gym = hip
    env = gym.make("hip/Pendulum-v1", num_envs=3)
    from gym_amd import error

    with pytest.raises(gym.error.ResetNeeded):                     # caught by the reference's type ...
        env.step(np.zeros((3, 1), np.float32))
    with pytest.raises(error.ResetNeeded):                         # ... and by the engine's
        env.step(np.zeros((3, 1), np.float32))
    env.reset(seed=0)
    env.step_async(np.zeros((3, 1), np.float32))
    with pytest.raises(gym.error.AlreadyPendingCallError):
        env.step_async(np.zeros((3, 1), np.float32))
    env.step_wait()
    with pytest.raises(gym.error.NoAsyncCallError):
        env.step_wait()
    with pytest.raises(gym.error.Error):                           # seeding.py:21-22
        env.reset(seed=-1)
    env.close()
    with pytest.raises(gym.error.ClosedEnvironmentError):
        env.reset()
    with pytest.raises(gym.error.UnregisteredEnv):
        from gym_amd.registration import spec

        spec("LunarLander-v2")


def test_the_references_wrappers_run_on_top_of_the_engine(hip):
    gym = hip
    n = 8
    env = gym.make("hip/CartPole-v1", num_envs=n, time_limit=12)

    class Counting(gym.vector.VectorEnvWrapper):                   # asserts isinstance(env, VectorEnv) (vector_env.py:289)
        calls = 0

        def step_wait(self):
            type(self).calls += 1
            return self.env.step_wait()

    wrapped = gym.wrappers.RecordEpisodeStatistics(Counting(env), deque_size=50)
    assert wrapped.num_envs == n and wrapped.is_vector_env
    obs, _ = wrapped.reset(seed=4)
    total, lengths = np.zeros(n), np.zeros(n, np.int64)
    seen = 0
    for t in range(40):
        a = env.action_space.sample()
        obs, rew, term, trunc, infos = wrapped.step(a)
        total += rew
        lengths += 1
        done = np.asarray(term) | np.asarray(trunc)     # the reference's wrapper hands the flags back as lists
        if done.any():
            assert np.array_equal(infos["_episode"], done) and np.array_equal(infos["_final_observation"], done)
            assert np.allclose(infos["episode"]["r"][done], total[done]) and np.array_equal(infos["episode"]["l"][done], lengths[done])
            assert all(infos["final_observation"][i].shape == (4,) for i in np.flatnonzero(done))
            assert all(infos["final_observation"][i] is None for i in np.flatnonzero(~done))
            total[done], lengths[done] = 0, 0
            seen += int(done.sum())
    assert Counting.calls == 40 and seen >= 2 * n and wrapped.episode_count == seen and len(wrapped.return_queue) == min(seen, 50)
    # VectorListInfo of the reference on top of that
    listed = gym.wrappers.VectorListInfo(gym.wrappers.RecordEpisodeStatistics(gym.make("hip/CartPole-v1", num_envs=4, time_limit=3)))
    listed.reset(seed=0)
    for _ in range(3):
        _, _, term, trunc, infos = listed.step(np.zeros(4, np.int64))
    assert isinstance(infos, list) and len(infos) == 4 and all("episode" in d and "final_observation" in d for d in infos)
    # ClipAction checks isinstance(env.action_space, Box) (clip_action.py:28)
    clipped = gym.wrappers.ClipAction(gym.make("hip/Pendulum-v1", num_envs=2))
    clipped.reset(seed=0)
    obs, rew, *_ = clipped.step(np.array([[5.0], [-5.0]], np.float32))
    assert obs.shape == (2, 3) and np.all(rew <= 0)


def test_reference_typed_env_pickles_as_the_engine_class(hip, monkeypatch):
    gym = hip
    env = gym.make("hip/MountainCar-v0", num_envs=2)
    env.reset(seed=0)
    monkeypatch.setattr(type(env).__mro__[1], "__getstate__", lambda self: {k: v for k, v in self.__dict__.items() if k != "_handle"})
    monkeypatch.setattr(type(env).__mro__[1], "__setstate__", lambda self, d: self.__dict__.update(d))
    clone = pickle.loads(pickle.dumps(env))
    assert isinstance(clone, gym.vector.VectorEnv) and type(clone).__mro__[1].__module__ == "gym_amd.vector_env"
    assert isinstance(clone.action_space, gym.spaces.MultiDiscrete) and clone.spec.id == "hip/MountainCar-v0"


def test_core_stays_usable_without_the_reference():
    """as_reference_env is a no-op when gym cannot be imported; nothing under gym_amd imports gym at module import time."""
    import subprocess
    import sys

    code = ("import sys; sys.modules['gym'] = None\n"
            "import gym_amd, gym_amd.vector_env, gym_amd.interop as I, gym_amd.spaces as S\n"
            "assert I.reference() is None\n"
            "class E: observation_space = S.Discrete(2)\n"
            "e = E(); assert I.as_reference_env(e) is e and type(e) is E\n"
            "print('ok')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=__import__("os").path.dirname(__import__("os").path.dirname(__file__)))
    assert out.stdout.strip() == "ok", out.stderr[-800:]


def test_the_references_normalisation_and_transform_wrappers_run_on_the_engines_vector_envs(hip, monkeypatch):
    """gym.wrappers.NormalizeObservation / NormalizeReward / TransformObservation / TransformReward / FlattenObservation of THE REFERENCE
    applied to what gym.make("hip/<id>", num_envs=n) returns — the vector-level use of those wrappers — and their numbers: the reference's
    NormalizeObservation over the engine equals the same wrapper over arrays of the same observations (it is the reference's own code
    running; what is checked is that the engine hands it what it needs: is_vector_env, num_envs, single_observation_space, the dtypes)."""
    gym = hip
    from gym.wrappers import (ClipAction, FlattenObservation, NormalizeObservation, NormalizeReward, RecordEpisodeStatistics, TransformObservation,
                              TransformReward)
    from gym.wrappers.normalize import RunningMeanStd
    from oracle_engine import FakeBlackjack, FakeTab

    from gym_amd import _native

    monkeypatch.setattr(_native, "Tab", FakeTab)
    monkeypatch.setattr(_native, "Blackjack", FakeBlackjack)
    n = 6
    plain, env = gym.make("hip/CartPole-v1", num_envs=n), NormalizeObservation(gym.make("hip/CartPole-v1", num_envs=n))
    rms = RunningMeanStd(shape=(4,))
    o0, o1 = plain.reset(seed=3)[0], env.reset(seed=3)[0]
    rms.update(o0)
    assert o1.dtype == np.float64 and np.array_equal(o1, (o0 - rms.mean) / np.sqrt(rms.var + 1e-8))
    for _ in range(30):
        a = plain.action_space.sample()
        x, y = plain.step(a)[0], env.step(a)[0]
        rms.update(x)
        assert np.array_equal(y, (x - rms.mean) / np.sqrt(rms.var + 1e-8))
    # the continuous-control recipe stacked with the reference's own classes at the vector level
    stack = TransformReward(NormalizeReward(TransformObservation(NormalizeObservation(ClipAction(RecordEpisodeStatistics(
        gym.make("hip/Pendulum-v1", num_envs=n, time_limit=9)))), lambda o: np.clip(o, -10, 10))), lambda r: np.clip(r, -10, 10))
    stack.reset(seed=0)
    ends = 0
    for _ in range(30):
        obs, rew, term, trunc, infos = stack.step(stack.action_space.sample())
        assert obs.shape == (n, 3) and obs.dtype == np.float64 and rew.shape == (n,) and np.abs(obs).max() <= 10
        ends += int("episode" in infos)
    assert ends == 3
    assert FlattenObservation(gym.make("hip/CartPole-v1", num_envs=n)).reset(seed=0)[0].shape == (n * 4,)      # (the batch space flattened: what it does to a SyncVectorEnv too)
    # ... and over the toy_text vector envs
    lake = NormalizeReward(RecordEpisodeStatistics(gym.make("hip/FrozenLake-v1", num_envs=n)))
    lake.reset(seed=0)
    for _ in range(40):
        obs, rew, term, trunc, infos = lake.step(lake.action_space.sample())
    assert obs.dtype == np.int64 and rew.dtype == np.float64 and lake.episode_count > 0
