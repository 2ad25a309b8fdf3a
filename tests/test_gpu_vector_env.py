"""HipVectorEnv (the gym.vector.SyncVectorEnv drop-in) on a real MI355X: the reference's own vector-env tests,
restated for this adapter — return contract (tests/vector/test_sync_vector_env.py:24-76), get/set_attr on "gravity"
(79-112), final_observation / final_info layout (tests/vector/test_vector_env_info.py:14-56,
tests/vector/test_vector_env.py:75-125), TimeLimit truncation (tests/wrappers/test_time_limit.py:17-57), determinism
under the same seed and actions (tests/envs/test_envs.py:63-115), reset bounds (tests/envs/test_env_implementation.py:
150-215), error behaviour — plus numeric parity of every returned value against the oracle."""
import numpy as np
from helpers import reference_wrapper_stub
import pytest

from helpers import ENV_IDS, GYM_IDS, LIMITS, MAX_OBS_ULPS, ulps32

pytestmark = pytest.mark.gpu

ALL_IDS = ["CartPole-v1", "CartPole-v0", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0"]


def _make(env_id, n, **kw):
    import gym_amd

    return gym_amd.make(env_id, num_envs=n, asynchronous=False, **kw)


@pytest.mark.parametrize("env_id", ALL_IDS)
def test_reset_and_step_contract(env_id):
    from gym_amd.spaces import Box, Discrete, MultiDiscrete

    n = 8
    env = _make(env_id, n)
    obs, infos = env.reset(seed=0)
    assert isinstance(env.observation_space, Box) and isinstance(obs, np.ndarray) and infos == {}
    assert obs.dtype == env.observation_space.dtype == np.float32
    assert obs.shape == (n,) + env.single_observation_space.shape == env.observation_space.shape
    assert env.observation_space.contains(obs)
    if isinstance(env.single_action_space, Discrete):
        assert isinstance(env.action_space, MultiDiscrete)
        actions = [env.single_action_space.sample() for _ in range(n)]  # list of python ints, like the reference's test
    else:
        actions = env.action_space.sample()
        assert actions.shape == (n, 1) and actions.dtype == np.float32
    obs2, rew, term, trunc, infos = env.step(actions)
    assert obs2.dtype == np.float32 and obs2.shape == obs.shape and obs2 is not obs
    assert isinstance(rew, np.ndarray) and rew.dtype == np.float64 and rew.shape == (n,)
    assert term.dtype == np.bool_ and term.shape == (n,) and trunc.dtype == np.bool_ and trunc.shape == (n,)
    assert isinstance(infos, dict)
    env.close()
    assert env.closed
    from gym_amd import error
    with pytest.raises(error.ClosedEnvironmentError):
        env.reset()


def test_call_answers_the_read_only_names_of_the_reference_and_refuses_methods():
    """VectorEnv.call(name) (sync_vector_env.py:171-190) returns the attribute of every sub-env: besides the physics attributes the
    device engine answers the names a learner reads off the reference's wrapped sub-envs — state, _elapsed_steps, _max_episode_steps,
    spec, render_mode, the single spaces — with the reference's container types; sub-env METHODS do not exist here and say so."""
    env = _make("CartPole-v1", 3)
    obs, _ = env.reset(seed=5)
    st = env.call("state")
    assert isinstance(st, tuple) and len(st) == 3 and all(isinstance(s, np.ndarray) and s.dtype == np.float64 for s in st)   # cartpole.py:202
    assert np.array_equal(np.asarray(st, dtype=np.float32), obs)
    obs1 = env.step(env.action_space.sample())[0]
    st1 = env.call("state")                                       # after a step: a tuple of floats (cartpole.py:160)
    assert all(isinstance(s, tuple) and len(s) == 4 and isinstance(s[0], float) for s in st1)
    assert np.array_equal(np.asarray(st1, dtype=np.float32), obs1)
    assert env.call("_elapsed_steps") == (1, 1, 1) and env.call("_max_episode_steps") == (500,) * 3
    assert env.call("render_mode") == (None,) * 3 and env.call("spec")[0].id == "CartPole-v1"
    assert env.call("action_space")[0] == env.single_action_space and env.get_attr("observation_space")[2] == env.single_observation_space
    for name, args in (("step", (0,)), ("reset", ()), ("render", ()), ("gravity", (1,))):
        with pytest.raises(NotImplementedError):
            env.call(name, *args)
    env.close()
    mcc = _make("MountainCarContinuous-v0", 2)
    o, _ = mcc.reset(seed=1)
    s2 = mcc.call("state")                                        # float64 right after reset (:182), float32 after a step (:171)
    assert all(isinstance(s, np.ndarray) and s.dtype == np.float64 and s.shape == (2,) for s in s2) and np.array_equal(np.stack(s2).astype(np.float32), o)
    o = mcc.step(mcc.action_space.sample())[0]
    s2 = mcc.call("state")
    assert all(s.dtype == np.float32 for s in s2) and np.array_equal(np.stack(s2), o)
    mcc.close()


def test_call_get_attr_set_attr_gravity():
    env = _make("CartPole-v1", 4)
    env.reset(seed=1)
    g = env.call("gravity")
    assert isinstance(g, tuple) and len(g) == 4 and all(isinstance(x, float) and x == 9.8 for x in g)
    env.set_attr("gravity", 3.72)
    assert env.get_attr("gravity") == (3.72,) * 4
    env.set_attr("gravity", [9.81] * 4)
    assert env.get_attr("gravity") == (9.81,) * 4
    with pytest.raises(ValueError):
        env.set_attr("gravity", [9.81, 1.62])  # wrong length (sync_vector_env.py:206-211)
    with pytest.raises(AttributeError):
        env.get_attr("no_such_attribute")
    assert env.get_attr("kinematics_integrator") == ("euler",) * 4
    # the new gravity is what the kernel integrates with
    from oracle.oracle import OracleVecEnv

    st, el = env.handle.get_state()
    o = OracleVecEnv(0, 4, 500)
    o.P[0] = 9.81
    o.state[:], o.elapsed[:] = st, el
    a = np.array([1, 0, 1, 1])
    obs, *_ = env.step(a)
    robs, *_ = o.step(a)
    assert ulps32(obs, robs).max() <= MAX_OBS_ULPS
    env.close()


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1", "Acrobot-v1"])
def test_pickle_round_trip_continues_identically(env_id):
    """tests/envs/test_envs.py:192-200 (test_pickle_env) for the vector env: the unpickled copy resets and steps like the
    original — here for 60 steps across autoresets, bit for bit, with a short TimeLimit so episodes end."""
    import pickle

    env = _make(env_id, 6, max_episode_steps=7)
    env.reset(seed=11)
    env.action_space.seed(5)
    if env_id == "CartPole-v1":
        env.set_attr("gravity", [9.8, 9.7, 9.6, 9.5, 9.4, 9.3])  # per-env attributes travel too
    for _ in range(9):
        env.step(env.action_space.sample())
    twin = pickle.loads(pickle.dumps(env))
    assert twin.get_attr("gravity" if env_id == "CartPole-v1" else ("g" if env_id == "Pendulum-v1" else "LINK_MASS_1")) == \
        env.get_attr("gravity" if env_id == "CartPole-v1" else ("g" if env_id == "Pendulum-v1" else "LINK_MASS_1"))
    for step in range(60):
        if step == 20:   # an unseeded reset continues the same streams in both
            o0, _ = env.reset()
            o1, _ = twin.reset()
            assert np.array_equal(o0, o1)
        a = env.action_space.sample()
        assert np.array_equal(a, twin.action_space.sample())   # the spaces' generators were pickled with the env
        r0, r1 = env.step(a), twin.step(a)
        for x, y in zip(r0[:4], r1[:4]):
            assert np.array_equal(x, y)
        assert set(r0[4].keys()) == set(r1[4].keys())
        if "final_observation" in r0[4]:
            for u, v in zip(r0[4]["final_observation"], r1[4]["final_observation"]):
                assert (u is None and v is None) or np.array_equal(u, v)
    env.close()
    twin.close()


def test_device_rollout_stream_ordering_helpers():
    """Actions computed on torch's current stream are complete before step() reads them, and ready() orders the outputs for
    the current stream: no host synchronisation anywhere, results equal the host-synchronised run."""
    import torch

    from gym_amd.rollout import DeviceRollout

    n = 1 << 18
    outs = []
    for synced in (True, False):
        r = DeviceRollout("CartPole-v1", n, seed=1, action_seed=2)
        r.reset(seed=1)
        r.synchronize()
        acc = torch.zeros(n, dtype=torch.float64, device="cuda")
        g = torch.Generator(device="cuda").manual_seed(0)
        for _ in range(30):
            # a "policy" on the current stream: enough work that the tensor is not ready when step() is called
            logits = torch.rand((n, 8), generator=g, device="cuda").cumsum(1)[:, -1]
            actions = (logits > 4.0).to(torch.int64)
            if synced:
                torch.cuda.synchronize()
            obs, rew, term, trunc = r.step(actions, want_final=False)
            if synced:
                r.synchronize()
            else:
                r.ready()
            acc += obs[:, 0].to(torch.float64) + rew
        torch.cuda.synchronize()
        outs.append((acc.clone(), r.handle.get_state()[0]))
        r.close()
    assert torch.equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_example_policy_search_learns_on_device():
    """examples/linear_policy_search.py end to end: policy on the caller's stream, env on the engine's, no host syncs in the
    loop; random linear policies average ~20-40 steps, two cross-entropy iterations must lift the population well above."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "linear_policy_search.py")
    spec = importlib.util.spec_from_file_location("linear_policy_search", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    history, w = mod.search(num_envs=4096, iterations=3, horizon=300, verbose=False)
    assert history[0][0] < 120 and history[-1][0] > 2 * history[0][0] and history[-1][1] >= 250, history
    assert w[2] > 0 and w[3] > 0    # push the cart towards the side the pole falls to


def test_example_dropin_loop_runs_on_the_engine():
    """examples/dropin_sync_vector_env.py: one loop written against the gym.vector interface (BASELINE.json configs[0]); here driven by
    gym_amd.vector.make.  A random CartPole policy lasts ~22 steps per episode."""
    import importlib.util
    import os

    import gym_amd

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "dropin_sync_vector_env.py")
    spec = importlib.util.spec_from_file_location("dropin_sync_vector_env", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.run(gym_amd.vector.make, "CartPole-v1", 8, 1000)
    assert r["episodes"] > 200 and 15 < r["mean_episode_length"] < 30, r
    r = mod.run(gym_amd.vector.make, "MountainCar-v0", 64, 450)
    assert r["episodes"] == 128 and r["mean_episode_length"] == 200, r       # a random policy never reaches the flag: two truncations per env
    # the continuous-control PPO wrapper list around every sub-env, written with gym_amd.wrappers' names (8192 sub-envs: the device kernels)
    r = mod.run(gym_amd.vector.make, "Pendulum-v1", 8192, 210, mod.ppo_recipe(gym_amd.wrappers))
    assert r["episodes"] == 8192 and r["mean_episode_length"] == 200, r


def test_device_rollout_state_dict_resumes_bit_identically():
    """Checkpoint / resume of the device-resident API incl. the fused episode statistics' running returns."""
    import pickle

    import torch

    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout("CartPole-v1", 3000, seed=3, action_seed=4)
    r.enable_episode_stats()
    r.reset(seed=3)
    r.rollout_per_step(40)
    snap = pickle.loads(pickle.dumps(r.state_dict()))
    out = r.rollout_per_step(64)
    r.synchronize()   # the engine runs on its own stream: finish before cloning on torch's current stream
    a = {k: v.clone() for k, v in out.items()}
    fresh = DeviceRollout("CartPole-v1", 3000, seed=99, action_seed=98)
    fresh.load_state_dict(snap)
    b = fresh.rollout_per_step(64)
    fresh.synchronize()
    assert set(a) == set(b)
    done = (a["terminated"] | a["truncated"]).bool()
    for k in a:
        if k in ("ep_return", "ep_length"):   # written only where an episode ended
            assert torch.equal(a[k][done], b[k][done]), k
        else:
            assert torch.equal(a[k], b[k]), k
    assert done.any() and "ep_return" in a
    with pytest.raises(ValueError):
        DeviceRollout("CartPole-v1", 2999).load_state_dict(snap)
    r.close()
    fresh.close()


def test_vector_env_wrapper_like_the_reference_tests():
    """tests/vector/test_vector_env_wrapper.py:7-31 with gym_amd's VectorEnvWrapper around the engine's envs."""
    import gym_amd

    class DummyWrapper(gym_amd.VectorEnvWrapper):
        def __init__(self, env):
            self.env = env
            self.counter = 0

        def reset_async(self, **kwargs):
            super().reset_async()
            self.counter += 1

    wrapped = DummyWrapper(gym_amd.make("FrozenLake-v1", asynchronous=False))
    wrapped.reset()
    assert wrapped.counter == 1
    wrapped.close()

    env = _make("CartPole-v1", 3)
    wrapped = DummyWrapper(_make("CartPole-v1", 3))
    assert np.allclose(wrapped.call("gravity"), env.call("gravity"))
    env.set_attr("gravity", [20.0, 20.0, 20.0])
    wrapped.set_attr("gravity", [20.0, 20.0, 20.0])
    assert np.allclose(wrapped.get_attr("gravity"), env.get_attr("gravity"))
    # a wrapped env steps exactly like the bare one
    o0, _ = env.reset(seed=3)
    o1, _ = wrapped.reset(seed=3)
    assert wrapped.counter == 1 and np.array_equal(o0, o1)
    a = np.array([1, 0, 1])
    for x, y in zip(env.step(a)[:4], wrapped.step(a)[:4]):
        assert np.array_equal(x, y)
    assert wrapped.unwrapped is wrapped.env and wrapped.num_envs == 3
    env.close()
    wrapped.close()
    assert wrapped.env.closed


@pytest.mark.parametrize("name", ["CartPole", "Pendulum", "Acrobot", "MountainCar", "MountainCarContinuous"])
def test_adapter_rollout_matches_oracle_with_final_observation(name):
    """Every value HipVectorEnv returns over a rollout with many episode ends (short TimeLimit), against the oracle
    stepping from the engine's own pre-step state with the same actions."""
    from oracle.oracle import OracleVecEnv

    n, limit, steps = 64, 9, 40
    env = _make(GYM_IDS[name], n, max_episode_steps=limit)
    env.reset(seed=3)
    env.action_space.seed(3)
    o = OracleVecEnv(ENV_IDS[name], n, limit)
    saw_final = 0
    for t in range(steps):
        st, el = env.handle.get_state()
        o.state[:], o.elapsed[:] = st, el
        a = env.action_space.sample()
        if name == "Pendulum" and t % 3 == 0:
            a = a * 2  # out-of-bounds torques are clipped by the env, not rejected (tests/envs/test_action_dim_check.py:90-136)
        obs, rew, term, trunc, infos = env.step(a)
        robs, rrew, rterm, rtrunc, rfin, rmask = o.step(a.reshape(n) if a.ndim == 2 else a)
        assert np.array_equal(term, rterm) and np.array_equal(trunc, rtrunc)
        np.testing.assert_allclose(rew, rrew, rtol=1e-13, atol=1e-9)
        done = term | trunc
        assert ulps32(obs[~done], robs[~done]).max(initial=0) <= MAX_OBS_ULPS
        assert np.all(trunc == (el + 1 >= limit))
        if done.any():
            # layout of vector_env.py:208-258: object arrays with None holes + boolean masks
            assert set(infos) == {"final_observation", "_final_observation", "final_info", "_final_info"}
            assert np.array_equal(infos["_final_observation"], done) and np.array_equal(infos["_final_info"], done)
            fo, fi = infos["final_observation"], infos["final_info"]
            assert fo.dtype == object and fo.shape == (n,) and fi.dtype == object
            for i in range(n):
                if done[i]:
                    assert fo[i].dtype == np.float32 and ulps32(fo[i], rfin[i]).max() <= MAX_OBS_ULPS and fi[i] == {}
                    assert env.single_observation_space.contains(obs[i])  # the returned row is the post-reset obs
                else:
                    assert fo[i] is None and fi[i] is None
            saw_final += int(done.sum())
        else:
            assert "final_observation" not in infos
    assert saw_final >= n * (steps // limit)
    env.close()


def test_time_limit_truncation_and_coinciding_termination():
    env = _make("CartPole-v1", 16, max_episode_steps=3)
    env.reset(seed=0)
    for k in range(1, 7):
        _, _, term, trunc, _ = env.step(np.zeros(16, dtype=np.int64))
        assert np.all(trunc == (k % 3 == 0))
    # terminated and truncated can coincide (tests/wrappers/test_time_limit.py:38-57)
    st, el = env.handle.get_state()
    st[0, :] = 2.399
    st[1, :] = 5.0
    el[:] = 2
    env.handle.set_state(st, el)
    _, _, term, trunc, infos = env.step(np.ones(16, dtype=np.int64))
    assert term.all() and trunc.all() and infos["_final_observation"].all()
    env.close()


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1", "MountainCarContinuous-v0"])
def test_determinism_same_seed_same_actions(env_id):
    n = 32
    a, b = _make(env_id, n), _make(env_id, n)
    oa, _ = a.reset(seed=42)
    ob, _ = b.reset(seed=42)
    assert np.array_equal(oa, ob)
    a.action_space.seed(42), b.action_space.seed(42)
    for _ in range(60):
        ra = a.step(a.action_space.sample())
        rb = b.step(b.action_space.sample())
        for x, y in zip(ra[:4], rb[:4]):
            assert np.array_equal(x, y)
    oc, _ = a.reset(seed=[7] * n)   # list of seeds: identical seeds give identical envs (sync_vector_env.py:102-110)
    assert np.all(oc == oc[0])
    od, _ = a.reset(seed=43)
    assert not np.array_equal(od, oa) and len(np.unique(od[:, 0])) == n
    a.close(), b.close()


def test_reset_options_bounds():
    env = _make("CartPole-v1", 256)
    obs, _ = env.reset(seed=1)
    assert np.all(np.abs(obs) <= 0.05)
    obs, _ = env.reset(seed=1, options={"low": -0.02, "high": 0.01})
    assert obs.min() >= -0.02 and obs.max() <= 0.01 and obs.min() < -0.015 and obs.max() > 0.005
    with pytest.raises(ValueError):
        env.reset(options={"low": 0.1, "high": -0.1})      # classic_control/utils.py:41-44
    with pytest.raises(ValueError):
        env.reset(options={"low": "abc"})                   # classic_control/utils.py:8-14
    env.close()
    env = _make("Pendulum-v1", 256)
    obs, _ = env.reset(seed=1, options={"x_init": 0.1, "y_init": 0.2})
    th = np.arctan2(obs[:, 1], obs[:, 0])
    assert np.all(np.abs(th) <= 0.1 + 1e-6) and np.all(np.abs(obs[:, 2]) <= 0.2 + 1e-6)
    env.close()
    env = _make("MountainCar-v0", 256)
    obs, _ = env.reset(seed=1)
    assert np.all((obs[:, 0] >= -0.6) & (obs[:, 0] <= -0.4)) and np.all(obs[:, 1] == 0)
    env.close()
    env = _make("Acrobot-v1", 64)
    obs, _ = env.reset(seed=1)
    st, _ = env.handle.get_state()
    assert np.array_equal(st, st.astype(np.float32).astype(np.float64)) and np.all(np.abs(st) <= 0.1 + 1e-7)  # acrobot.py:188-190
    env.close()


def test_error_behaviour():
    from gym_amd import error

    env = _make("CartPole-v1", 8)
    with pytest.raises(error.ResetNeeded):
        env.step(np.zeros(8, dtype=np.int64))        # order_enforcing.py:33-37
    with pytest.raises(error.Error):
        env.reset(seed=-1)                           # seeding.py:21-22
    env.reset(seed=0)
    with pytest.raises(AssertionError):
        env.step(np.full(8, 2, dtype=np.int64))      # cartpole.py:131-132
    with pytest.raises(AssertionError):
        env.step(np.zeros(8, dtype=np.float32))      # Discrete.contains rejects floats
    with pytest.raises(error.NoAsyncCallError):
        env.step_wait()
    env.step_async(np.zeros(8, dtype=np.int64))
    with pytest.raises(error.AlreadyPendingCallError):
        env.step_async(np.zeros(8, dtype=np.int64))
    env.step_wait()
    env.step(np.ones(8, dtype=np.int64))             # still usable after the errors
    env.close()
    with pytest.raises(error.UnregisteredEnv):
        _make("LunarLander-v2", 8)                   # Box2D: out of scope
    with pytest.raises(TypeError):
        _make("CartPole-v1", 8, g=1.0)               # not a CartPole kwarg
    env = _make("Pendulum-v1", 4, g=9.81)            # pendulum.py:91
    assert env.get_attr("g") == (9.81,) * 4
    env.close()


def test_random_policy_episode_length_cartpole():
    """P3 distributional check: mean random-policy CartPole episode ~22 steps."""
    env = _make("CartPole-v1", 4096)
    env.reset(seed=9)
    env.action_space.seed(9)
    done_count, steps = 0, 120
    for _ in range(steps):
        _, _, term, trunc, _ = env.step(env.action_space.sample())
        done_count += int((term | trunc).sum())
    mean_len = 4096 * steps / done_count
    assert 19.0 < mean_len < 25.5, mean_len
    env.close()


def test_set_attr_per_env_values_like_the_reference_test():
    """tests/vector/test_sync_vector_env.py:101-110: set_attr("gravity", [9.81, 3.72, 8.87, 1.62]) gives every sub-env
    its own value; every sub-env then integrates with ITS gravity (checked against one oracle per distinct value)."""
    from oracle.oracle import OracleVecEnv

    env = _make("CartPole-v1", 4)
    env.reset(seed=5)
    env.set_attr("gravity", [9.81, 3.72, 8.87, 1.62])
    assert env.get_attr("gravity") == (9.81, 3.72, 8.87, 1.62)
    assert env.get_attr("force_mag") == (10.0,) * 4
    env.set_attr("kinematics_integrator", ["euler", "semi-implicit", "euler", "semi-implicit"])
    assert env.call("kinematics_integrator") == ("euler", "semi-implicit", "euler", "semi-implicit")
    for _ in range(25):
        st, el = env.handle.get_state()
        a = env.action_space.sample()
        obs, rew, term, trunc, _ = env.step(a)
        for i, (g, ki) in enumerate(zip((9.81, 3.72, 8.87, 1.62), (0.0, 1.0, 0.0, 1.0))):
            o = OracleVecEnv(0, 1, 500)
            o.P[0], o.P[10] = g, ki
            o.state[:, 0], o.elapsed[:] = st[:, i], el[i]
            robs, _, rterm, rtrunc, _, _ = o.step(a[i:i + 1])
            assert rterm[0] == term[i] and rtrunc[0] == trunc[i]
            if not (term[i] or trunc[i]):
                assert ulps32(obs[i], robs[0]).max() <= MAX_OBS_ULPS, (i, obs[i], robs[0])
    # equal values again -> broadcast mode, fast kernels
    env.set_attr("gravity", [9.8] * 4)
    env.set_attr("kinematics_integrator", "euler")
    assert env.get_attr("gravity") == (9.8,) * 4 and not env._per_env
    env.step(env.action_space.sample())
    env.close()
    # sampled fused rollouts keep working while attributes differ (they take the per-step kernel)
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout("Pendulum-v1", 64, seed=1, action_seed=2)
    tbl = r.handle.get_params_per_env()
    tbl[3] = np.linspace(9.0, 11.0, 64)  # g
    r.handle.set_params_per_env(tbl)
    r.reset(seed=1)
    out = r.rollout_per_step(10, mode="fused")
    r.synchronize()
    assert np.isfinite(out["obs"].cpu().numpy()).all()
    r.close()


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1"])
@pytest.mark.parametrize("n", [8, 20000, 50000])
def test_copy_false_and_zero_copy_equal_the_copying_path(env_id, n):
    """SyncVectorEnv(copy=False) (sync_vector_env.py:61-63,163): the returned observations are the internal buffer — here a
    view of the pinned, device-mapped I/O block the kernel writes over PCIe.  Same numbers as copy=True, step for step; the
    observation array is reused, rewards/flags are fresh unless zero_copy=True; views stay valid after close().  Above 2 MiB of
    step I/O (n = 50000) every mode returns arrays the caller owns (views of a pooled pinned block, one DMA per step)."""
    import gym_amd

    envs = [gym_amd.make(env_id, n), gym_amd.make(env_id, n, copy=False), gym_amd.make(env_id, n, zero_copy=True)]
    outs = [e.reset(seed=3)[0] for e in envs]
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    envs[0].action_space.seed(1)
    prev_obs_id = outs[1].__array_interface__["data"][0]
    shared = not envs[1]._packed
    assert shared == (n <= 20000)
    ndone = 0
    for t in range(30):
        a = envs[0].action_space.sample()
        res = [e.step(a) for e in envs]
        for r in res[1:]:
            for x, y in zip(res[0][:4], r[:4]):
                assert x.dtype == y.dtype and np.array_equal(x, y)
        if shared:
            assert res[1][0].__array_interface__["data"][0] == prev_obs_id      # the same buffer every step
        done = res[0][2] | res[0][3]
        if done.any():
            for r in res[1:]:
                fo = r[4]["final_observation"]
                ref = res[0][4]["final_observation"]
                assert all(np.array_equal(fo[i], ref[i]) for i in np.flatnonzero(done))
            ndone += int(done.sum())
    if env_id == "CartPole-v1":
        assert ndone > 0
    rew_copy, rew_view = res[1][1], res[2][1]
    kept = [(x, x.copy()) for r in res[1:] for x in r[:4]]
    envs[1].step(a)
    envs[2].step(a)
    if shared:
        assert rew_copy is not envs[1]._io()["reward"]                          # copy=False: rewards are copies
        assert rew_view.__array_interface__["data"][0] == envs[2]._io()["reward"].__array_interface__["data"][0]
    else:
        assert all(np.array_equal(x, y) for x, y in kept)                       # owned arrays: the next step left them alone
    obs_view = res[2][0]
    for e in envs:
        e.close()
    assert np.isfinite(obs_view).all() and obs_view.shape == (n, outs[0].shape[1])   # still readable after close()


@pytest.mark.parametrize("mode", ["copy", "copy_false", "zero_copy"])
@pytest.mark.parametrize("limit", [3, 40])
def test_large_env_final_observations_travel_packed(mode, limit):
    """Above 2 MiB of step I/O the adapter no longer copies the dense final_obs array over PCIe: the device packs (index, row)
    pairs of the envs that finished (final_count_kernel + final_pack_kernel) and the library scatters them on the host.  TimeLimit 3: a third of
    the 200 000 envs finish per step — more than the speculative first transfer holds (n / 8), so the second one runs;
    TimeLimit 40: a few percent.  Every info["final_observation"] row must be the terminal observation the oracle computes, for
    the copying adapter (pooled arrays) and for the views of the pinned block alike; arrays the caller keeps stay intact."""
    from oracle.oracle import OracleVecEnv

    n, steps = 200_000, 7
    kw = {"copy": {}, "copy_false": dict(copy=False), "zero_copy": dict(zero_copy=True)}[mode]
    env = _make("CartPole-v1", n, max_episode_steps=limit, **kw)
    env.reset(seed=11)
    env.action_space.seed(12)
    o = OracleVecEnv(ENV_IDS["CartPole"], n, limit)
    kept = []
    for t in range(steps):
        st, el = env.handle.get_state()
        o.state[:], o.elapsed[:] = st, el
        a = env.action_space.sample()
        obs, rew, term, trunc, infos = env.step(a)
        robs, rrew, rterm, rtrunc, rfin, rmask = o.step(a)
        assert np.array_equal(term, rterm) and np.array_equal(trunc, rtrunc)
        done = term | trunc
        if mode == "copy":
            kept.append((obs, obs.copy(), rew, rew.copy()))     # a fresh array per call: later steps must not touch these
        if not done.any():
            assert not infos
            continue
        assert np.array_equal(infos["_final_observation"], done) and np.array_equal(infos["_final_info"], done)
        fo = infos["final_observation"]
        idx = np.flatnonzero(done)
        assert idx.size > (n // 8 if limit == 3 and (t + 1) % 3 == 0 else 0)
        got = np.stack([fo[i] for i in idx])
        assert ulps32(got, rfin[idx]).max() <= MAX_OBS_ULPS
        assert all(fo[i] is None for i in np.flatnonzero(~done)[:1000])
    for o1, o2, r1, r2 in kept:
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2)
    assert len({id(k[0]) for k in kept}) == len(kept)
    env.close()


@pytest.mark.parametrize("name", ["CartPole", "Pendulum", "Acrobot", "MountainCar", "MountainCarContinuous"])
@pytest.mark.parametrize("n,packed", [(64, False), (150_000, False), (150_000, True)])
def test_one_dma_block_step_equals_per_array_step(name, n, packed):
    """mxv_step_host_block (one pooled pinned block per step, a single DMA) against mxv_step_host (four copies into separate
    arrays) on twin handles: identical outputs; packed final rows arrive in ascending env order (= np.flatnonzero of the done
    mask) and equal the dense rows; a block is never handed out again while the caller holds an array of it, and is once dropped."""
    from gym_amd import _native

    kind = getattr(_native, name.upper().replace("MOUNTAINCARCONTINUOUS", "MOUNTAINCAR_CONT"))
    limit = 9
    h = _native.Handle(kind, n, limit, seed=3, action_seed=4)
    g = _native.Handle(kind, n, limit, seed=3, action_seed=4)
    h.reset_host()
    g.reset_host()
    if packed:
        assert h.final_packed(True)
    rng = np.random.default_rng(5)
    held, seen_ptrs, ndone = [], set(), 0
    for t in range(24):
        a = rng.integers(0, h.NA, n) if h.NA > 0 else rng.uniform(-2.5, 2.5, n).astype(np.float32)
        obs, rew, term, trunc, fin = h.step_host_block(a, want_final=True)
        o2, r2, te2, tr2, f2 = g.step_host(a, want_final=True)
        assert np.array_equal(obs, o2) and np.array_equal(rew, r2) and np.array_equal(term, te2) and np.array_equal(trunc, tr2)
        assert obs.dtype == np.float32 and rew.dtype == r2.dtype and term.dtype == np.bool_
        done = term | trunc
        ndone += int(done.sum())
        if packed:
            assert fin is None
            idx, rows = h.final_packed_rows()
            assert np.array_equal(idx, np.flatnonzero(done))
            assert np.array_equal(rows, f2[idx])
        else:
            assert np.array_equal(fin[done], f2[done])
        if t < 3:
            held.append((obs, obs.copy(), rew, rew.copy(), term, term.copy()))
        seen_ptrs.add(obs.ctypes.data)
    assert ndone >= 2 * n              # TimeLimit 9: every env finished at least twice in 24 steps
    for o1, o1c, r1, r1c, t1, t1c in held:
        assert np.array_equal(o1, o1c) and np.array_equal(r1, r1c) and np.array_equal(t1, t1c)
    assert len({x[0].ctypes.data for x in held}) == 3
    assert len(seen_ptrs) <= 6          # blocks are recycled (3 held + the ones in flight), not allocated per step
    del held, obs, rew, term, trunc, fin
    h.close()
    g.close()


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Acrobot-v1", "MountainCar-v0"])
def test_large_env_discrete_action_dtypes_and_out_of_range_values(env_id):
    """int64 and int32 actions of a large env give the same trajectories, and an out-of-range value anywhere — negative,
    2^40 + 1 (a valid low byte: catches any narrowing of the upload), NA — raises Discrete.contains' AssertionError with the
    engine usable afterwards."""
    import gym_amd
    from gym_amd import _native

    n = 1 << 17
    a = _native.Handle(getattr(_native, {"CartPole-v1": "CARTPOLE", "Acrobot-v1": "ACROBOT", "MountainCar-v0": "MOUNTAINCAR"}[env_id]),
                       n, 30, seed=3, action_seed=4)
    b = _native.Handle(a.env_id, n, 30, seed=3, action_seed=4, flags=_native.FLAG_ACTION_I32)
    a.reset_host(), b.reset_host()
    rng = np.random.default_rng(0)
    for t in range(40):
        act = rng.integers(0, a.NA, n)
        ra = a.step_host_block(act, want_final=False)
        rb = b.step_host_block(act.astype(np.int32), want_final=False)
        for x, y in zip(ra[:4], rb[:4]):
            assert np.array_equal(x, y)
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    a.close(), b.close()
    env = gym_amd.make(env_id, num_envs=n)
    env.reset(seed=0)
    good = np.zeros(n, dtype=np.int64)
    for bad_value in (-1, (1 << 40) + 1, env.single_action_space.n):
        bad = good.copy()
        bad[n - 7] = bad_value
        with pytest.raises(AssertionError):
            env.step(bad)
        env.step(good)
    env.close()


def test_snapshots_carry_a_format_version_and_old_ones_are_refused_with_a_reason():
    """Handle.snapshot() is versioned: a snapshot written before the reset stream was indexed by per-env reset ordinals (no `episodes`,
    no `format`) cannot continue bit-identically and is refused with a message, not a KeyError."""
    from gym_amd import _native

    h = _native.Handle(_native.CARTPOLE, 64, 500, seed=1, action_seed=2)
    h.reset_host()
    snap = h.snapshot()
    assert snap["format"] == _native.SNAPSHOT_FORMAT == 2 and snap["episodes"].shape == (64,)
    h.restore(snap)
    old = {k: v for k, v in snap.items() if k not in ("format", "episodes")}
    with pytest.raises(ValueError, match="snapshot format 1"):
        h.restore(old)
    # written after the ordinals came in but before the `format` key did: semantically format 2, and restored as such
    keyless = {k: v for k, v in snap.items() if k != "format"}
    h.step_host(np.zeros(64, dtype=np.int64))
    h.restore(keyless)
    assert np.array_equal(h.get_state()[0], snap["state"]) and np.array_equal(h.get_episodes(), snap["episodes"])
    h.close()


def test_vector_make_with_recognised_sub_env_wrappers_against_the_reference():
    """gym.vector.make(id, n, wrappers=[partial(TimeLimit, max_episode_steps=12), RecordEpisodeStatistics]) (gym/vector/__init__.py:56-65;
    the reference's tests/vector/test_vector_make.py applies a wrapper to every sub-env the same way): the two wrappers the engine owns an
    equivalent of are mapped onto it — the TimeLimit counter with min(12, 500), the fused episode accumulators — and what a caller sees is
    what the reference's per-sub-env wrappers produce (tests/golden/vector_make_wrappers_CartPole.npz, made by running the reference):
    masks exactly, and `infos["final_info"][i]["episode"]` = {"r": float32, "l": int32, "t": float} for the sub-envs whose episode ended,
    nothing at the vector level."""
    import functools
    import os

    import gym_amd
    from gym_amd.wrappers import RecordEpisodeStatistics, SubEnvEpisodeStatistics

    TimeLimit = reference_wrapper_stub("TimeLimit")            # recognised by name: the reference's gym.wrappers.TimeLimit is not importable on the GPU box

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vector_make_wrappers_CartPole.npz"))
    T, N = g["action"].shape
    env = gym_amd.make("CartPole-v1", num_envs=N, wrappers=[functools.partial(TimeLimit, max_episode_steps=int(g["max_episode_steps"])),
                                                             RecordEpisodeStatistics])
    assert isinstance(env, SubEnvEpisodeStatistics) and env.get_attr("_max_episode_steps") == (12,) * N
    env.reset(seed=1)
    handle = env.unwrapped.handle
    episodes = 0
    for t in range(T):
        handle.set_state(np.ascontiguousarray(g["state_pre"][t].T), g["elapsed_pre"][t])
        obs, rew, term, trunc, infos = env.step(g["action"][t])
        assert np.array_equal(term, g["terminated"][t]) and np.array_equal(trunc, g["truncated"][t]), t
        assert np.array_equal(rew, g["reward"][t]) and "episode" not in infos and "_episode" not in infos
        done = g["ep_mask"][t]
        nd = ~done
        assert ulps32(obs[nd], g["obs"][t][nd]).max() <= MAX_OBS_ULPS if nd.any() else True
        if done.any():
            assert np.array_equal(infos["_final_info"], done)
            for i in np.flatnonzero(done):
                ep = infos["final_info"][i]["episode"]
                assert isinstance(ep["r"], np.float32) and isinstance(ep["l"], np.int32) and isinstance(ep["t"], float)
                assert ep["r"] == g["ep_r"][t][i] and ep["l"] == g["ep_l"][t][i], (t, i, ep)
                episodes += 1
            assert all(infos["final_info"][i] is None for i in np.flatnonzero(nd))
        else:
            assert "final_info" not in infos
    assert episodes == int(g["ep_mask"].sum()) == 59
    env.close()
    # a bare TimeLimit class (max_episode_steps=None -> the spec's own limit, time_limit.py:40-43) changes nothing
    plain = gym_amd.make("CartPole-v1", num_envs=4, wrappers=TimeLimit)
    assert plain.get_attr("_max_episode_steps") == (500,) * 4
    plain.close()
    shorter = gym_amd.make("CartPole-v1", num_envs=4, max_episode_steps=5, wrappers=[functools.partial(TimeLimit, max_episode_steps=9)])
    assert shorter.get_attr("_max_episode_steps") == (5,) * 4             # whichever TimeLimit is shorter ends the episode
    shorter.close()
    lake = gym_amd.make("FrozenLake-v1", num_envs=4, wrappers=functools.partial(TimeLimit, max_episode_steps=6))
    assert lake.get_attr("_max_episode_steps") == (6,) * 4
    lake.close()


def test_reset_outputs_are_ordered_on_the_callers_stream():
    """`obs = r.reset(seed=s)` of the device-resident front-ends returns a tensor the CALLER may read at once on its own stream: the engines
    launch on a non-blocking stream of their own, so without an ordering the caller's `.cpu()` could run before the reset kernel — or, as
    here, before the long rollout queued in front of it (round 5: tools/soak.py read a Blackjack reset before it had happened)."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from gym_amd.toy_text import BlackjackRollout, TabularRollout
    from oracle.oracle import OracleBlackjack, OracleVecEnv

    n = 1 << 18
    r = DeviceRollout("CartPole-v1", n, seed=3, action_seed=4)
    r.reset(seed=3)
    out = r.trajectory_buffers(256, layout="separate")
    o = OracleVecEnv(0, n, 500, seed=77, action_seed=4)
    want = o.reset(seed=77)
    for _ in range(3):
        r.rollout_per_step(256, out=out)                     # ~0.4 ms of work queued on the engine's stream ...
        got = r.reset(seed=77).cpu().numpy()                 # ... then the reset, read at once on the default stream
        assert np.array_equal(got, want)
        r.handle.seed(3)
    r.close()
    b = BlackjackRollout(n, seed=1, action_seed=2)
    b.reset(seed=1)
    traj = b.trajectory_buffers(128, layout="separate")
    wantb = OracleBlackjack(n, seed=9, action_seed=2, sab=True).reset(seed=9)
    for _ in range(3):
        b.rollout_per_step(128, out=traj)
        assert np.array_equal(b.reset(seed=9).cpu().numpy(), wantb)
    b.close()
    t = TabularRollout("FrozenLake8x8-v1", n, seed=1, action_seed=2)
    first = t.reset(seed=5).cpu().numpy().copy()
    tr = t.trajectory_buffers(128, layout="separate")
    for _ in range(3):
        t.rollout_per_step(128, out=tr)
        assert np.array_equal(t.reset(seed=5).cpu().numpy(), first) and (first == 0).all()      # FrozenLake starts in state 0
    t.close()


@pytest.mark.parametrize("name", ["CartPole", "Pendulum"])
def test_vector_make_normalize_wrappers_against_the_reference(name):
    """The GPU twin of tests/test_host_logic.py::test_vector_make_normalize_wrappers_replay_the_reference_bit_for_bit: the same golden (the
    reference run with TimeLimit + NormalizeObservation + NormalizeReward + RecordEpisodeStatistics around every sub-env) through the HIP
    engine — masks and episode lengths exact, normalised observations / rewards / episode returns within the engine's tolerances."""
    from helpers import replay_vector_make_normalize

    assert replay_vector_make_normalize(name, exact=False) > 50


def test_vector_make_clipaction_charges_the_clipped_action_on_the_device():
    """wrappers=ClipAction is not an identity for MountainCarContinuous-v0 — the reward's `action[0] ** 2 * 0.1`
    (continuous_mountain_car.py:169) must see the clipped action.  The reference's own run with actions up to +-6 (golden), on the GPU."""
    from helpers import replay_vector_make_clipaction

    assert replay_vector_make_clipaction(exact=False) == 60


@pytest.mark.parametrize("device", [None, True])
def test_vector_make_transform_wrappers_on_the_device(device):
    from helpers import replay_vector_make_transform

    assert replay_vector_make_transform(exact=False, device=device) > 40


@pytest.mark.parametrize("name", ["Pendulum", "MountainCarContinuous"])
def test_vector_make_rescaleaction_on_the_device(name):
    from helpers import replay_vector_make_rescaleaction

    assert replay_vector_make_rescaleaction(name, exact=False) == 60
