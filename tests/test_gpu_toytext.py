"""SURVEY.md §8(f)-4 on the MI355X: the table-driven toy_text engine (mxv_tab_*, through the C ABI).

  (a) the reference's SyncVectorEnv trajectories replayed with the reference's own recorded uniforms: BIT-EXACT
      (observations, rewards, flags, infos["prob"] incl. the _add_info dtype quirk, final_observation / final_info);
  (b) Philox mode against the oracle twin (sampled actions, transition and reset streams): bit-exact, fused K-step
      launch == K single-step launches == action-tape replay;
  (c) the HipTabularVectorEnv surface (dtypes, infos, errors) — what gym.vector.SyncVectorEnv returns for these ids;
  (d) 2^20 envs: shard invariance, rewind-and-replay, MDP consistency of every recorded transition.
"""
import numpy as np
import pytest

from helpers import TOYTEXT_CASES, load_toytext_golden, replay_toytext, toytext_mdp

pytestmark = pytest.mark.gpu


def _tab(mdp, n, limit, **kw):
    from gym_amd import _native

    return _native.Tab(mdp.num_states, mdp.num_actions, mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated,
                       mdp.initial_cum, n, limit, **kw)


class _HipAdapter:
    def __init__(self, mdp, n, limit):
        self.h = _tab(mdp, n, limit)

    def set_state(self, state, elapsed):
        self.h.set_state(state, elapsed)

    def step(self, actions, uniforms):
        obs, rew, term, trunc, prob, fin, fprob = self.h.step_host(actions, uniforms)
        return dict(obs=obs, reward=rew, terminated=term, truncated=trunc, prob=prob, final_obs=fin, final_prob=fprob)


@pytest.mark.parametrize("tag", TOYTEXT_CASES)
def test_device_replays_reference_trajectories_bit_exact(tag):
    g = load_toytext_golden(tag)
    ndone = replay_toytext(g, _HipAdapter)
    assert ndone == int(g["final_mask"].sum()) and ndone > 0


@pytest.mark.parametrize("gid,limit", [("FrozenLake-v1", 100), ("FrozenLake8x8-v1", 9), ("Taxi-v3", 13), ("CliffWalking-v0", None)])
def test_philox_mode_equals_oracle_twin_and_fused_equals_single_steps(gid, limit):
    import torch
    from gym_amd.toy_text import TabularRollout
    from oracle.oracle import OracleTabEnv

    n, K = 3001, 50
    runs = {}
    for mode in ("fused", "single", "tape"):
        r = TabularRollout(gid, n, seed=31, action_seed=32, max_episode_steps=limit)
        obs0 = r.reset(seed=31).cpu().numpy().copy()
        if mode == "fused":
            out = r.rollout_per_step(K)
        elif mode == "single":
            out = r.trajectory_buffers(K)
            for k in range(K):
                r.handle.rollout(1, out["obs"][k], out["reward"][k], out["terminated"][k], out["truncated"][k], out["prob"][k],
                                 actions_out_dev=out["actions"][k], per_step=False)
        else:
            out = r.rollout_tape(torch.from_numpy(runs["fused"][1]["actions"]).cuda())
        r.synchronize()
        from gym_amd import _native
        assert r.handle.last_kernel() == (_native.TAB_KERNEL_TRAJECTORY if mode == "fused" else _native.TAB_KERNEL_GENERAL), mode
        runs[mode] = (obs0, {k: v.cpu().numpy() for k, v in out.items()}, r.handle.get_state())
        r.close()
    for mode in ("single", "tape"):
        assert np.array_equal(runs["fused"][0], runs[mode][0])
        for k, v in runs["fused"][1].items():
            assert np.array_equal(v, runs[mode][1][k]), (mode, k)
        assert all(np.array_equal(a, b) for a, b in zip(runs["fused"][2], runs[mode][2]))
    from gym_amd.toy_text import TOY_TEXT_REGISTRY
    mdp = TOY_TEXT_REGISTRY[gid].build()
    orc = OracleTabEnv(mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated, mdp.initial_cum, n, limit,
                       seed=31, action_seed=32)
    assert np.array_equal(orc.reset(seed=31), runs["fused"][0])
    dev = runs["fused"][1]
    ndone = 0
    for k in range(K):
        o = orc.step()
        for name in ("actions", "obs", "reward", "prob"):
            assert np.array_equal(o[name], dev[name][k]), (k, name)
        assert np.array_equal(o["terminated"], dev["terminated"][k].astype(bool))
        assert np.array_equal(o["truncated"], dev["truncated"][k].astype(bool))
        ndone += int(o["final_mask"].sum())
    assert np.array_equal(orc.state, runs["fused"][2][0]) and np.array_equal(orc.elapsed, runs["fused"][2][1])
    if limit is not None:
        assert ndone > 0


def test_hip_tabular_vector_env_contract():
    import gym_amd
    from gym_amd import error
    from gym_amd.spaces import Discrete, MultiDiscrete
    from gym_amd.toy_text import HipTabularVectorEnv

    env = gym_amd.make("FrozenLake-v1", 8, max_episode_steps=5)
    assert isinstance(env, HipTabularVectorEnv) and env.num_envs == 8 and env.is_vector_env
    assert isinstance(env.single_observation_space, Discrete) and env.single_observation_space.n == 16
    assert isinstance(env.observation_space, MultiDiscrete) and isinstance(env.action_space, MultiDiscrete)
    with pytest.raises(error.ResetNeeded):
        env.step(np.zeros(8, np.int64))
    obs, infos = env.reset(seed=7)
    assert obs.dtype == np.int64 and obs.shape == (8,) and np.all(obs == 0)           # the single start state
    assert infos["prob"].dtype == np.int64 and np.all(infos["prob"] == 1) and infos["_prob"].all()   # {"prob": 1}
    obs2, _ = env.reset(seed=7)
    assert np.array_equal(obs, obs2)
    seen_int = seen_float = False
    env.action_space.seed(0)
    for _ in range(40):
        obs, rew, term, trunc, infos = env.step(env.action_space.sample())
        assert obs.dtype == np.int64 and rew.dtype == np.float64 and term.dtype == np.bool_ and trunc.dtype == np.bool_
        done = term | trunc
        if done[0]:   # VectorEnv._add_info: sub-env 0's int 1 types the whole array
            assert infos["prob"].dtype == np.int64
            seen_int = True
        else:
            assert infos["prob"].dtype == np.float64
            seen_float = True
        assert np.all(np.asarray(infos["prob"])[done] == 1)
        if done.any():
            assert infos["final_observation"].dtype == np.int64 and np.array_equal(infos["_final_observation"], done)
            assert np.all(infos["final_observation"][~done] == 0)
            fi = infos["final_info"]
            assert fi.dtype == object and all((fi[i] is None) != bool(done[i]) for i in range(8))
            assert all(set(fi[i]) == {"prob"} and fi[i]["prob"] in (1.0, 1.0 / 3.0) for i in np.flatnonzero(done))
            assert np.all(obs[done] == 0)                                                # returned obs = the reset state
        else:
            assert "final_observation" not in infos
    assert seen_int and seen_float
    with pytest.raises(KeyError):
        env.step(np.array([0, 1, 2, 3, 4, 0, 0, 0]))
    P = env.get_attr("P")[0]
    assert P[0][0] == [(1.0 / 3.0, 0, 0.0, False), (1.0 / 3.0, 0, 0.0, False), (1.0 / 3.0, 4, 0.0, False)]
    env.close()
    with pytest.raises(error.ClosedEnvironmentError):
        env.reset()

    taxi = gym_amd.make("Taxi-v3", 5)
    obs, infos = taxi.reset(seed=[1, 2, 3, 4, 5])
    assert infos["prob"].dtype == np.float64                                             # {"prob": 1.0}
    am = infos["action_mask"]
    assert am.dtype == object and am[0].dtype == np.int8 and am[0].shape == (6,)
    assert np.array_equal(np.stack(list(am)), taxi.mdp.action_mask[obs])
    obs, rew, term, trunc, infos = taxi.step(np.array([4, 4, 5, 0, 1]))
    assert set(np.unique(rew)) <= {-1.0, -10.0, 20.0}
    assert np.array_equal(np.stack(list(infos["action_mask"])), taxi.mdp.action_mask[obs])
    taxi.close()
    with pytest.raises(TypeError):
        gym_amd.make("CliffWalking-v0", 4, is_slippery=False)
    det = gym_amd.make("FrozenLake-v1", 4, is_slippery=False)
    det.reset(seed=0)
    o, r, te, tr, _ = det.step(np.array([2, 2, 1, 1]))           # RIGHT -> 1, DOWN -> 4, deterministically
    assert np.array_equal(o, [1, 1, 4, 4])
    det.close()


def test_taxi_helpers_through_call():
    """TaxiEnv.encode / decode / action_mask (taxi.py:210-252) answered through VectorEnv.call like any sub-env method."""
    import gym_amd

    env = gym_amd.make("Taxi-v3", num_envs=3)
    obs, info = env.reset(seed=4)
    for s in obs.tolist():
        parts = [list(d) for d in env.call("decode", s)]
        assert parts[0] == parts[1] == parts[2] and env.call("encode", *parts[0]) == (s,) * 3
        masks = env.call("action_mask", s)
        assert len(masks) == 3 and masks[0].dtype == np.int8 and masks[0].shape == (6,)
    assert all(np.array_equal(info["action_mask"][i], env.call("action_mask", int(obs[i]))[0]) for i in range(3))
    with pytest.raises(AttributeError):
        env.call("no_such_method")
    env.close()


@pytest.mark.parametrize("gid", ["FrozenLake-v1", "Taxi-v3"])
def test_pickle_round_trip_continues_identically(gid):
    """tests/envs/test_envs.py:192-200 for the tabular vector envs: the unpickled copy steps like the original, bit for bit."""
    import pickle

    import gym_amd

    env = gym_amd.make(gid, num_envs=50, max_episode_steps=9)
    env.reset(seed=21)
    env.action_space.seed(2)
    for _ in range(7):
        env.step(env.action_space.sample())
    twin = pickle.loads(pickle.dumps(env))
    for step in range(40):
        if step == 15:
            assert np.array_equal(env.reset()[0], twin.reset()[0])
        a = env.action_space.sample()
        assert np.array_equal(a, twin.action_space.sample())
        r0, r1 = env.step(a), twin.step(a)
        for x, y in zip(r0[:4], r1[:4]):
            assert np.array_equal(x, y)
        assert np.array_equal(r0[4]["prob"], r1[4]["prob"]) and set(r0[4]) == set(r1[4])
    env.close()
    twin.close()


def test_full_size_properties():
    """2^20 FrozenLake8x8 envs, 64-step fused rollouts."""
    import torch
    from gym_amd.toy_text import TabularRollout

    n, K = 1 << 20, 64
    full = TabularRollout("FrozenLake8x8-v1", n, seed=9, action_seed=10)
    obs0 = full.reset(seed=9)
    full.synchronize()   # engine stream -> finish before cloning on torch's current stream
    obs0 = obs0.clone()
    a = full.rollout_per_step(K)
    full.synchronize()
    # shard invariance: the same logical vector env on 4 handles (global-index Philox streams)
    parts = []
    for w in range(4):
        sh = TabularRollout("FrozenLake8x8-v1", n // 4, env_offset=w * (n // 4), seed=9, action_seed=10)
        o0 = sh.reset(seed=9)
        sh.synchronize()
        o0 = o0.clone()
        parts.append((o0, sh.rollout_per_step(K)))
        sh.synchronize()
        sh.close()
    assert torch.equal(obs0, torch.cat([p[0] for p in parts]))
    for k in a:
        assert torch.equal(a[k], torch.cat([p[1][k] for p in parts], dim=1)), k
    # rewind and replay the recorded actions: identical trajectory (transition uniforms depend on (seed, env, t) only)
    full.handle.set_state(obs0.cpu().numpy().astype(np.int32), np.zeros(n, np.int32))
    full.handle.set_counters(0, 1)
    b = full.rollout_tape(a["actions"])
    full.synchronize()
    for k in ("obs", "reward", "terminated", "truncated", "prob"):
        assert torch.equal(a[k], b[k]), k
    # every recorded transition is a transition of the MDP
    mdp = full.mdp
    obs = a["obs"].cpu().numpy()
    act = a["actions"].cpu().numpy()
    done = (a["terminated"] | a["truncated"]).cpu().numpy().astype(bool)
    rew = a["reward"].cpu().numpy()
    prev = np.concatenate([obs0.cpu().numpy()[None], obs[:-1]])
    k = 17
    nxt_ok = np.zeros(n, bool)
    for i in range(mdp.max_transitions):
        nxt_ok |= (mdp.next_state[prev[k], act[k], i] == obs[k]) & (mdp.cum_prob[prev[k], act[k], i] >= 0)
    assert nxt_ok[~done[k]].all()
    assert np.all(obs[k][done[k]] == 0)                        # autoreset to the start state
    assert set(np.unique(rew)) <= {0.0, 1.0} and rew.sum() > 0  # some random walks reach the goal
    assert np.all(rew[~a["terminated"].cpu().numpy().astype(bool)] == 0)
    p = a["prob"].cpu().numpy()
    assert np.array_equal(p == 1.0, done) and np.all(np.isin(p, [1.0, 1.0 / 3.0]))   # reset()'s prob 1 vs slippery 1/3
    assert 0.005 < done.mean() < 0.3                           # random FrozenLake8x8 episodes last tens of steps
    full.close()


def test_large_table_falls_back_to_global_memory():
    """A custom 20x20 map (33 600 B of cum/prob/reward + ... > 64 KiB of LDS) runs from L2 instead; same semantics."""
    from gym_amd.toy_text import frozen_lake_mdp
    from oracle.oracle import OracleTabEnv

    rng = np.random.default_rng(0)
    rows = [["F"] * 20 for _ in range(20)]
    for _ in range(40):
        rows[rng.integers(20)][rng.integers(20)] = "H"
    rows[0][0], rows[19][19] = "S", "G"
    mdp = frozen_lake_mdp(desc=["".join(r) for r in rows])
    assert mdp.num_states * 4 * 3 * 28 + mdp.num_states * 8 > 64 * 1024
    n = 2048
    h = _tab(mdp, n, 50, seed=1, action_seed=2)
    orc = OracleTabEnv(mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated, mdp.initial_cum, n, 50, seed=1,
                       action_seed=2)
    assert np.array_equal(h.reset_host(), orc.reset())
    rng = np.random.default_rng(1)
    for _ in range(30):
        a = rng.integers(0, 4, n)
        obs, rew, term, trunc, prob, fin, fprob = h.step_host(a)
        o = orc.step(a)
        assert np.array_equal(obs, o["obs"]) and np.array_equal(rew, o["reward"]) and np.array_equal(term, o["terminated"])
        assert np.array_equal(trunc, o["truncated"]) and np.array_equal(prob, o["prob"])
    h.close()


@pytest.mark.parametrize("gid", ["FrozenLake8x8-v1", "Taxi-v3", "CliffWalking-v0"])
def test_compact_trajectories_hold_the_same_values(gid):
    """MXV_TAB_FLAG_COMPACT: int32 observations / actions and float32 rewards / probs on the trajectory tensors (18 instead of 34 bytes
    per env-step) — the same streams, the same values, for sampled rollouts and for tape-driven ones; state and counters identical."""
    import torch
    from gym_amd.toy_text import TabularRollout

    n, K = 70001, 64
    wide = TabularRollout(gid, n, seed=5, action_seed=6, max_episode_steps=19)
    comp = TabularRollout(gid, n, seed=5, action_seed=6, max_episode_steps=19, compact=True)
    wide.reset(seed=5), comp.reset(seed=5)
    for rep in range(2):
        a, b = wide.rollout_per_step(K), comp.rollout_per_step(K)
        wide.synchronize(), comp.synchronize()
        assert b["obs"].dtype == torch.int32 and b["actions"].dtype == torch.int32 and b["reward"].dtype == torch.float32 and b["prob"].dtype == torch.float32
        for key in ("obs", "actions", "terminated", "truncated"):
            assert torch.equal(a[key].to(torch.int64), b[key].to(torch.int64)), (gid, rep, key)
        for key in ("reward", "prob"):
            assert torch.equal(a[key].to(torch.float32), b[key]), (gid, rep, key)
        assert int((a["terminated"] | a["truncated"]).sum()) > 0
    tape = a["actions"].clone()
    c = wide.rollout_tape(tape)
    d = comp.rollout_tape(tape.to(torch.int32))
    wide.synchronize(), comp.synchronize()
    assert torch.equal(c["obs"].to(torch.int64), d["obs"].to(torch.int64)) and torch.equal(c["reward"].to(torch.float32), d["reward"])
    for x, y in zip(wide.handle.get_state(), comp.handle.get_state()):
        assert np.array_equal(x, y)
    wide.close(), comp.close()


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("gid,limit", [("FrozenLake-v1", 100), ("FrozenLake8x8-v1", 11), ("Taxi-v3", 7), ("CliffWalking-v0", 23),
                                       ("FrozenLake-v1:det", 9)])
def test_trajectory_kernel_equals_general_kernel(gid, limit, compact):
    """tab_traj_kernel (integer thresholds, packed table, DPP word transpose, lazily drawn reset uniforms) against tab_step_kernel on
    the same handle configuration: every output of every step, state and counters, over launches that start and end inside the
    four-step blocks of the action stream (K = 3, 50, 1, 64, 6) and a ragged last workgroup (n = 70 001); per-env seeds included."""
    import torch
    from gym_amd import _native
    from gym_amd.toy_text import TabularRollout

    kw = {"is_slippery": False} if gid.endswith(":det") else {}
    gid = gid.split(":")[0]
    n = 70001
    fast = TabularRollout(gid, n, seed=5, action_seed=6, max_episode_steps=limit, compact=compact, **kw)
    gen = TabularRollout(gid, n, seed=5, action_seed=6, max_episode_steps=limit, compact=compact, general_kernel=True, **kw)
    seeds = np.random.default_rng(3).integers(0, 2 ** 63, n, dtype=np.uint64)
    ended = 0
    for rnd, per_env in enumerate((False, True)):
        for r in (fast, gen):
            r.handle.seed(5 + rnd, seeds if per_env else None)
            r.reset()
        for K in (3, 50, 1, 64, 6):
            a, b = fast.rollout_per_step(K), gen.rollout_per_step(K)
            fast.synchronize(), gen.synchronize()
            assert fast.handle.last_kernel() == _native.TAB_KERNEL_TRAJECTORY and gen.handle.last_kernel() == _native.TAB_KERNEL_GENERAL
            for key in ("obs", "actions", "reward", "prob", "terminated", "truncated"):
                assert a[key].dtype == b[key].dtype and torch.equal(a[key], b[key]), (gid, compact, per_env, K, key)
            ended += int((a["terminated"] | a["truncated"]).sum())
        for x, y in zip(fast.handle.get_state(), gen.handle.get_state()):
            assert np.array_equal(x, y)
        assert fast.handle.get_counters() == gen.handle.get_counters()
    assert ended > n
    fast.close(), gen.close()


def test_mdps_the_packing_refuses_stay_on_the_general_kernel():
    """A transition list of length 2, a reward that is no float32 value, a cumulative sum that stops short of 1: the handle keeps the
    general kernel for trajectory launches too — and agrees with the CPU twin."""
    import torch
    from gym_amd import _native
    from gym_amd.toy_text import frozen_lake_mdp
    from oracle.oracle import OracleTabEnv

    base = frozen_lake_mdp(map_name="4x4")
    S, A, M = base.num_states, base.num_actions, base.max_transitions
    cases = []
    rew = np.array(base.reward, np.float64).copy(); rew[rew == 1.0] = 0.1
    cases.append(("reward 0.1", dict(reward=rew)))
    cum = np.array(base.cum_prob, np.float64).copy(); cum[cum == 1.0] = 1.0 - 2.0 ** -30
    cases.append(("cum stops short of 1", dict(cum_prob=cum)))
    cum2, prob2 = np.array(base.cum_prob)[:, :, :2].copy(), np.array(base.prob)[:, :, :2].copy()
    full = cum2[:, :, 1] >= 0
    cum2[:, :, 1][full] = 1.0; prob2[:, :, 1][full] = 2.0 / 3.0
    cases.append(("M = 2", dict(cum_prob=cum2, prob=prob2, next_state=np.array(base.next_state)[:, :, :2].copy(),
                                reward=np.array(base.reward)[:, :, :2].copy(), terminated=np.array(base.terminated)[:, :, :2].copy())))
    n, K = 5000, 40
    for what, over in cases:
        t = dict(cum_prob=base.cum_prob, prob=base.prob, next_state=base.next_state, reward=base.reward, terminated=base.terminated)
        t.update(over)
        h = _native.Tab(S, A, t["cum_prob"], t["prob"], t["next_state"], t["reward"], t["terminated"], base.initial_cum, n, 30, seed=1,
                        action_seed=2)
        orc = OracleTabEnv(t["cum_prob"], t["prob"], t["next_state"], t["reward"], t["terminated"], base.initial_cum, n, 30, seed=1,
                           action_seed=2)
        dev = torch.device("cuda", 0)
        out = {k: torch.empty((K, n), dtype=dt, device=dev) for k, dt in (("obs", torch.int64), ("reward", torch.float64), ("actions", torch.int64),
                                                                           ("prob", torch.float64), ("terminated", torch.uint8), ("truncated", torch.uint8))}
        torch.cuda.synchronize()
        assert np.array_equal(h.reset_host(), orc.reset())
        h.rollout(K, out["obs"], out["reward"], out["terminated"], out["truncated"], out["prob"], actions_out_dev=out["actions"], per_step=True)
        h.sync()
        assert h.last_kernel() == _native.TAB_KERNEL_GENERAL, what
        for k in range(K):
            o = orc.step()
            for name in ("actions", "obs", "reward", "prob"):
                assert np.array_equal(o[name], out[name][k].cpu().numpy()), (what, k, name)
            assert np.array_equal(o["terminated"], out["terminated"][k].cpu().numpy().astype(bool)), (what, k)
        h.close()


@pytest.mark.parametrize("S,A,M,point_mass", [(5, 3, 3, False), (7, 1, 3, True), (11, 5, 1, False), (3, 3, 1, True)])
def test_trajectory_kernel_on_synthetic_mdps(S, A, M, point_mass):
    """Tables no registered env has — odd S * A (the packed entries' alignment), unequal transition probabilities, lists of one, two and
    three entries side by side, an initial distribution with zero-probability states at both ends and M = 3 — through the trajectory
    kernel, the general kernel and the CPU twin."""
    import torch
    from gym_amd import _native
    from oracle.oracle import OracleTabEnv

    rng = np.random.default_rng(S * 100 + A * 10 + M)
    cum = np.full((S, A, M), -1.0)
    prob = np.zeros((S, A, M))
    for s in range(S):
        for a in range(A):
            n = 1 if M == 1 else int(rng.integers(1, M + 1))
            p = {1: [1.0], 2: [[0.25, 0.75], [0.5, 0.5], [0.875, 0.125]][int(rng.integers(3))],
                 3: [[0.25, 0.25, 0.5], [1.0 / 3.0] * 3, [0.125, 0.5, 0.375]][int(rng.integers(3))]}[n]
            prob[s, a, :n] = p
            cum[s, a, :n] = np.cumsum(p)
    nxt = rng.integers(0, S, (S, A, M)).astype(np.int32)
    rew = rng.integers(-3, 4, (S, A, M)).astype(np.float64) * 0.5
    term = (rng.random((S, A, M)) < 0.15).astype(np.uint8)
    if point_mass:
        init = np.zeros(S); init[S // 2] = 1.0
    else:
        init = np.zeros(S); k = rng.choice(np.arange(1, S - 1), size=min(3, S - 2), replace=False); init[k] = 1.0 / len(k)
    icum = np.cumsum(init)
    icum[np.flatnonzero(init)[-1]:] = 1.0
    n, K = 4097, 37
    dev = torch.device("cuda", 0)
    outs = []
    for general in (False, True):
        h = _native.Tab(S, A, cum, prob, nxt, rew, term, icum, n, 9, seed=3, action_seed=4, general_kernel=general)
        out = {k: torch.empty((K, n), dtype=dt, device=dev) for k, dt in (("obs", torch.int64), ("reward", torch.float64), ("actions", torch.int64),
                                                                           ("prob", torch.float64), ("terminated", torch.uint8), ("truncated", torch.uint8))}
        torch.cuda.synchronize()
        obs0 = h.reset_host()
        h.rollout(K, out["obs"], out["reward"], out["terminated"], out["truncated"], out["prob"], actions_out_dev=out["actions"], per_step=True)
        h.sync()
        assert h.last_kernel() == (_native.TAB_KERNEL_GENERAL if general else _native.TAB_KERNEL_TRAJECTORY)
        outs.append((obs0, {k: v.cpu().numpy() for k, v in out.items()}, h.get_state()))
        h.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k
    assert all(np.array_equal(x, y) for x, y in zip(outs[0][2], outs[1][2]))
    orc = OracleTabEnv(cum, prob, nxt, rew, term, icum, n, 9, seed=3, action_seed=4)
    assert np.array_equal(orc.reset(), outs[0][0])
    ended = 0
    for k in range(K):
        o = orc.step()
        for name in ("actions", "obs", "reward", "prob"):
            assert np.array_equal(o[name], outs[0][1][name][k]), (k, name)
        assert np.array_equal(o["terminated"], outs[0][1]["terminated"][k].astype(bool))
        assert np.array_equal(o["truncated"], outs[0][1]["truncated"][k].astype(bool))
        ended += int(o["final_mask"].sum())
    assert ended > n


def test_trajectory_kernel_randomised_launch_shapes():
    """24 random cases (env, batch size incl. ragged and sub-workgroup ones, TimeLimit, dtype set, a sequence of launch lengths that
    walks through every phase of the four-step action blocks, per-env seeds or not, a counter set far beyond 2^32): trajectory kernel ==
    general kernel on every output of every step, state and counters."""
    import torch
    from gym_amd import _native
    from gym_amd.toy_text import TabularRollout

    rng = np.random.default_rng(2024)
    gids = ["FrozenLake-v1", "FrozenLake8x8-v1", "Taxi-v3", "CliffWalking-v0"]
    for case in range(24):
        gid = gids[case % 4]
        n = int(rng.choice([1, 63, 64, 255, 256, 257, 1000, 4097, 20_000]))
        limit = int(rng.integers(2, 40))
        compact = bool(rng.integers(2))
        kw = dict(seed=int(rng.integers(1 << 30)), action_seed=int(rng.integers(1 << 30)), max_episode_steps=limit, compact=compact)
        fast, gen = TabularRollout(gid, n, **kw), TabularRollout(gid, n, general_kernel=True, **kw)
        seeds = rng.integers(0, 2 ** 63, n, dtype=np.uint64) if rng.integers(2) else None
        t0 = int(rng.integers(1 << 33, 1 << 40))
        for r in (fast, gen):
            r.handle.seed(kw["seed"], seeds)
            r.reset()
            if case % 5 == 0:
                r.handle.set_counters(t0, r.handle.get_counters()[1])
        for K in [int(k) for k in rng.integers(1, 23, size=5)]:
            a, b = fast.rollout_per_step(K), gen.rollout_per_step(K)
            fast.synchronize(), gen.synchronize()
            assert fast.handle.last_kernel() == _native.TAB_KERNEL_TRAJECTORY and gen.handle.last_kernel() == _native.TAB_KERNEL_GENERAL
            for key in ("obs", "actions", "reward", "prob", "terminated", "truncated"):
                assert torch.equal(a[key], b[key]), (case, gid, n, limit, compact, K, key)
        for x, y in zip(fast.handle.get_state(), gen.handle.get_state()):
            assert np.array_equal(x, y), (case, gid, n)
        assert fast.handle.get_counters() == gen.handle.get_counters()
        fast.close(), gen.close()
