"""The fused paths held against the ORACLE, not against the device's own stand-alone paths (VERDICT r4, weak #3 / Next #2).

* `rollout_kernel_v3<..., STATS = 1 | 2 | 3>` (the batch moments of NormalizeObservation / NormalizeReward formed by the launch that
  writes the trajectory, gym/wrappers/normalize.py:12-48,72-145): after >= 64 steps
    (a) end to end against the oracle TWIN — `OracleVecEnv` stepped with the same seeds produces its own trajectory (observations within 2
        float32 ulps of the device's, helpers.MAX_OBS_ULPS), `oracle.RunningNorm` (mode 1: the definition the kernels implement) normalises
        it; normalised observations / rewards and the running mean / var / count / returns agree within what 2 ulps of the inputs allow;
    (b) the oracle fed the DEVICE's trajectory tensors: the same numbers to 1e-9 (summation order of exact fp64 sums is all that differs),
        the running returns BIT-EXACT (the recurrence is IEEE arithmetic in the reference's order);
    (c) the reference's OWN arithmetic (oracle mode 0: float32 row-by-row batch moments — pinned bit-exact against the reference's wrappers
        by tests/golden/normalize_*.npz, tests/test_normalize_oracle.py) on the device's trajectory, within the reference's float32
        accumulation error (helpers.norm_obs_bound);
  all five env kinds, both dtype sets, ragged and sub-tile sizes.
* hipGraph replays (the device clock) against the oracle twins — `OracleVecEnv` for the recorded policy loop (`graphed_loop`) and for
  sampled steps, `OracleTabEnv` for the table engine's trajectory kernel, `OracleBlackjack` for Blackjack — not only against the same
  calls made one by one (tests/test_gpu_graph_capture.py)."""
import numpy as np
import pytest

from helpers import ENV_IDS, ENV_NAMES, GYM_IDS, LIMITS, MAX_OBS_ULPS, REWARD_ATOL, norm_obs_bound, ulps32

pytestmark = pytest.mark.gpu


def _oracle_trajectory(name, n, limit, seed, action_seed, steps):
    """The oracle twin's own trajectory: obs f32 [T][n][O], reward f64 [T][n], terminated / truncated bool [T][n], actions."""
    from oracle.oracle import OracleVecEnv

    o = OracleVecEnv(ENV_IDS[name], n, limit, seed=seed, action_seed=action_seed)
    o.reset(seed=seed)
    obs, rew, te, tr, act = [], [], [], [], []
    for _ in range(steps):
        a = o.sample_actions()
        ob, rw, t1, t2, _, _ = o.step(a)
        obs.append(ob), rew.append(rw), te.append(t1), tr.append(t2), act.append(a)
    return np.stack(obs), np.stack(rew), np.stack(te), np.stack(tr), np.stack(act)


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("name,n", [("CartPole", 33_000), ("CartPole", 130), ("Pendulum", 4_097), ("Acrobot", 2_500), ("MountainCar", 65),
                                    ("MountainCarContinuous", 6_000)])
def test_fused_moments_against_the_oracle(name, n, compact):
    import torch
    from gym_amd.rollout import DeviceRollout
    from oracle.oracle import RunningNorm

    K, launches, gamma = 24, 3, 0.97                  # 72 steps
    limit = min(LIMITS[name], 21)
    r = DeviceRollout(GYM_IDS[name], n, seed=12, action_seed=13, reward_f32=compact, action_i32=compact, max_episode_steps=limit)
    r.reset(seed=12)
    nz = r.make_normalizer(gamma=gamma)
    r.fuse_reward_normalizer(nz)
    out = r.trajectory_buffers(K, layout="separate", obs_partials=True, ret_partials=True)
    dev = {k: [] for k in ("obs", "reward", "terminated", "truncated", "actions", "y", "q")}
    for _ in range(launches):
        r.rollout_per_step(K, out=out)
        li = r.handle.last_launch()
        assert li["kernel"] == 1 and li["out_mode"] == (2 if compact else 1), li                   # the fused trajectory kernel, STATS = 3
        y = nz.normalize_obs(out["obs"], partials=out["obs_partials"])
        q = nz.normalize_rewards(out["reward"], out["terminated"], out["truncated"], partials=out["ret_partials"])
        r.synchronize()
        for k in ("obs", "reward", "terminated", "truncated", "actions"):
            dev[k].append(out[k].cpu().numpy().copy())
        dev["y"].append(y.cpu().numpy()), dev["q"].append(q.cpu().numpy())
    dev = {k: np.concatenate(v) for k, v in dev.items()}
    T = K * launches
    mean, var, count = nz.backend.obs_state()
    rmean, rvar, rcount, returns = nz.backend.reward_state()
    r.close()

    # (a) the oracle twin, end to end
    o_obs, o_rew, o_te, o_tr, o_act = _oracle_trajectory(name, n, limit, 12, 13, T)
    assert np.array_equal(dev["terminated"].astype(bool), o_te) and np.array_equal(dev["truncated"].astype(bool), o_tr)
    assert np.array_equal(dev["actions"].astype(o_act.dtype), o_act) if o_act.dtype.kind == "i" else np.array_equal(dev["actions"], o_act)
    assert ulps32(dev["obs"], o_obs).max() <= MAX_OBS_ULPS
    twin = RunningNorm(n, o_obs.shape[2], gamma=gamma, mode=1)
    y_twin = twin.normalize_obs(o_obs)
    q_twin = twin.normalize_rewards(o_rew, o_te, o_tr)
    # 2 float32 ulps of x move y by 2 ulp(x) / std; the moments move by less than that
    std = np.sqrt(twin.obs_var + 1e-8)
    tol_y = (MAX_OBS_ULPS + 1) * np.spacing(np.abs(o_obs).astype(np.float32)).astype(np.float64) / std + 1e-6 * (1.0 + np.abs(y_twin))
    assert np.all(np.abs(dev["y"] - y_twin) <= tol_y), float((np.abs(dev["y"] - y_twin) / tol_y).max())
    ra = REWARD_ATOL.get(name, 0.0)
    q_tol = (2e-6 if compact else 1e-9) * np.abs(q_twin) + (ra * 50 + 1e-12)                       # (compact: float32 rewards in and out)
    assert np.all(np.abs(dev["q"].astype(np.float64) - q_twin) <= q_tol)
    np.testing.assert_allclose(mean, twin.obs_mean, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(var, twin.obs_var, rtol=1e-5)
    assert count == twin.obs_count[0] == pytest.approx(1e-4 + T * n, rel=1e-12)
    np.testing.assert_allclose(returns, twin.returns, rtol=1e-6 if compact else 1e-10, atol=ra * 50 + 1e-12)
    np.testing.assert_allclose([rmean[0], rvar[0]], [twin.ret_mean[0], twin.ret_var[0]], rtol=1e-5 if compact else 1e-8)
    assert rcount == twin.ret_count[0]

    # (b) the oracle on the device's own trajectory: only the order of exact fp64 sums differs; the running returns are bit-identical
    same = RunningNorm(n, o_obs.shape[2], gamma=gamma, mode=1)
    np.testing.assert_allclose(dev["y"], same.normalize_obs(dev["obs"]), rtol=1e-9, atol=1e-12)
    q_same = same.normalize_rewards(dev["reward"].astype(np.float64), dev["terminated"], dev["truncated"])
    np.testing.assert_allclose(dev["q"].astype(np.float64), q_same, rtol=2e-6 if compact else 1e-11, atol=0)
    assert np.array_equal(returns, same.returns) and (returns != 0).any()
    np.testing.assert_allclose(mean, same.obs_mean, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(var, same.obs_var, rtol=1e-11)
    assert rvar[0] == pytest.approx(same.ret_var[0], rel=1e-11)

    # (c) the reference's own arithmetic (float32 batch moments), within its float32 accumulation error
    ref = RunningNorm(n, o_obs.shape[2], gamma=gamma, mode=0)
    y_ref = ref.normalize_obs(dev["obs"])
    err, bound = np.abs(dev["y"] - y_ref), norm_obs_bound(dev["obs"], y_ref)
    assert np.all(err <= bound), float((err / bound).max())
    q_ref = ref.normalize_rewards(dev["reward"].astype(np.float64), dev["terminated"], dev["truncated"])
    np.testing.assert_allclose(dev["q"].astype(np.float64), q_ref, rtol=2e-6 if compact else 1e-10, atol=0)
    assert int((dev["terminated"] | dev["truncated"]).sum()) > 0


@pytest.mark.parametrize("stats", ["obs", "returns"])
def test_single_moment_instantiations_against_the_oracle(stats):
    """STATS = 1 (observation moments only) and STATS = 2 (discounted returns only), CartPole at a ragged size."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from oracle.oracle import RunningNorm

    n, K, launches, gamma = 9_000, 32, 2, 0.99
    r = DeviceRollout("CartPole-v1", n, seed=3, action_seed=4, max_episode_steps=18)
    r.reset(seed=3)
    nz = r.make_normalizer(gamma=gamma)
    if stats == "returns":
        r.fuse_reward_normalizer(nz)
    out = r.trajectory_buffers(K, layout="separate", obs_partials=stats == "obs", ret_partials=stats == "returns")
    ys, obs, rew, te, tr = [], [], [], [], []
    for _ in range(launches):
        r.rollout_per_step(K, out=out)
        y = (nz.normalize_obs(out["obs"], partials=out["obs_partials"]) if stats == "obs" else
             nz.normalize_rewards(out["reward"], out["terminated"], out["truncated"], partials=out["ret_partials"]))
        r.synchronize()
        ys.append(y.cpu().numpy()), obs.append(out["obs"].cpu().numpy().copy()), rew.append(out["reward"].cpu().numpy().copy())
        te.append(out["terminated"].cpu().numpy().copy()), tr.append(out["truncated"].cpu().numpy().copy())
    ys, obs, rew, te, tr = (np.concatenate(v) for v in (ys, obs, rew, te, tr))
    o_obs, o_rew, o_te, o_tr, _ = _oracle_trajectory("CartPole", n, 18, 3, 4, K * launches)
    assert np.array_equal(te.astype(bool), o_te) and np.array_equal(tr.astype(bool), o_tr) and ulps32(obs, o_obs).max() <= MAX_OBS_ULPS
    twin = RunningNorm(n, 4, gamma=gamma, mode=1)
    if stats == "obs":
        np.testing.assert_allclose(ys, twin.normalize_obs(o_obs), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(nz.backend.obs_state()[0], twin.obs_mean, rtol=1e-6, atol=1e-7)
    else:
        np.testing.assert_allclose(ys, twin.normalize_rewards(o_rew, o_te, o_tr), rtol=1e-10)     # CartPole's rewards are exactly 1.0
        assert np.array_equal(nz.backend.reward_state()[3], twin.returns)
    r.close()


# ---- hipGraph replays against the oracle twins --------------------------------------------------------------------------------------
def test_graphed_policy_loop_against_the_oracle():
    """DeviceRollout.graphed_loop (policy + step recorded once, replayed): the oracle twin stepped with the actions the recorded policy
    chose reproduces every replayed step — masks exactly, observations within 2 float32 ulps — and so do sampled steps in a graph."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from oracle.oracle import OracleVecEnv

    n, K, replays = 2048, 16, 4
    torch.manual_seed(1)
    W = torch.randn(4, 2, device="cuda")
    policy = lambda obs: (obs @ W).argmax(dim=1)   # noqa: E731
    a = DeviceRollout("CartPole-v1", n, seed=11, action_seed=12, max_episode_steps=20)
    obs0 = a.reset(seed=11).cpu().numpy().copy()
    traj = {"obs": torch.empty((K, n, 4), device="cuda"), "act": torch.empty((K, n), dtype=torch.int64, device="cuda"),
            "term": torch.empty((K, n), dtype=torch.uint8, device="cuda"), "trunc": torch.empty((K, n), dtype=torch.uint8, device="cuda")}
    last_act = {}

    def pol(obs):
        last_act["a"] = policy(obs)
        return last_act["a"]

    def record(k):
        traj["obs"][k].copy_(a.obs), traj["act"][k].copy_(last_act["a"]), traj["term"][k].copy_(a.terminated), traj["trunc"][k].copy_(a.truncated)

    with torch.cuda.stream(a.stream):         # the matrix product's library initialises lazily, and may not do so inside a capture
        policy(a.obs)
        a.stream.synchronize()
    g = a.graphed_loop(pol, K, warmup=0, on_step=record)
    o = OracleVecEnv(0, n, 20, seed=11, action_seed=12)
    assert np.array_equal(o.reset(seed=11), obs0)
    ended = 0
    for _ in range(replays):
        g.replay()
        a.synchronize()
        got = {k: v.cpu().numpy() for k, v in traj.items()}
        for k in range(K):
            ob, _, te, tr, _, _ = o.step(got["act"][k])
            assert np.array_equal(te, got["term"][k].astype(bool)) and np.array_equal(tr, got["trunc"][k].astype(bool)), k
            assert ulps32(got["obs"][k], ob).max() <= MAX_OBS_ULPS, k
            ended += int(te.sum() + tr.sum())
    assert ended > 0 and a.handle.get_counters()[0] == replays * K
    a.close()


@pytest.mark.parametrize("kind", ["CartPole", "Pendulum", "MountainCar"])
def test_sampled_steps_in_a_graph_against_the_oracle(kind):
    import torch
    from gym_amd.rollout import DeviceRollout

    n, per_graph, replays = 4096, 8, 6
    limit = min(LIMITS[kind], 30)
    r = DeviceRollout(GYM_IDS[kind], n, seed=3, action_seed=4, max_episode_steps=limit)
    r.reset(seed=3)
    with torch.cuda.stream(r.stream):
        r.enable_graph_capture()
        bufs = {"obs": torch.empty((per_graph, n, r.O), device=r.device), "terminated": torch.empty((per_graph, n), dtype=torch.uint8, device=r.device),
                "truncated": torch.empty((per_graph, n), dtype=torch.uint8, device=r.device),
                "actions": torch.empty((per_graph, n), dtype=r.action_dtype, device=r.device)}
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=r.stream):
            for k in range(per_graph):
                r.step_sampled(record_actions=True)
                bufs["obs"][k].copy_(r.obs), bufs["terminated"][k].copy_(r.terminated), bufs["truncated"][k].copy_(r.truncated)
                bufs["actions"][k].copy_(r.actions.reshape(n))
        got = []
        for _ in range(replays):
            g.replay()
            got.append({k: v.clone() for k, v in bufs.items()})
    r.synchronize()
    dev = {k: torch.cat([d[k] for d in got]).cpu().numpy() for k in bufs}
    r.close()
    o_obs, _, o_te, o_tr, o_act = _oracle_trajectory(kind, n, limit, 3, 4, per_graph * replays)
    assert np.array_equal(dev["actions"], o_act.astype(dev["actions"].dtype)) if o_act.dtype.kind == "i" else np.array_equal(dev["actions"], o_act)
    assert np.array_equal(dev["terminated"].astype(bool), o_te) and np.array_equal(dev["truncated"].astype(bool), o_tr)
    assert ulps32(dev["obs"], o_obs).max() <= MAX_OBS_ULPS and int(o_te.sum() + o_tr.sum()) > 0


@pytest.mark.parametrize("gid,compact", [("FrozenLake8x8-v1", False), ("Taxi-v3", True)])
def test_tabular_graph_replays_against_the_oracle(gid, compact):
    """The table engine's trajectory kernel recorded in a caller's graph (device clock) vs OracleTabEnv: bit-exact, replay after replay."""
    import torch
    from gym_amd.toy_text import TabularRollout
    from oracle.oracle import OracleTabEnv

    n, K, replays = 8192, 6, 7
    r = TabularRollout(gid, n, seed=3, action_seed=4, max_episode_steps=9, compact=compact)
    obs0 = r.reset(seed=3).cpu().numpy().copy()
    out = r.trajectory_buffers(K, layout="separate")
    log = []
    with torch.cuda.stream(r.stream):
        r.handle.set_device_clock(True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=r.stream):
            r.rollout_per_step(K, out=out)
        for _ in range(replays):
            g.replay()
            log.append({k: v.clone() for k, v in out.items()})
    r.synchronize()
    m = r.mdp
    o = OracleTabEnv(m.cum_prob, m.prob, m.next_state, m.reward, m.terminated, m.initial_cum, n, 9, seed=3, action_seed=4)
    assert np.array_equal(o.reset(seed=3), obs0)
    ended = 0
    for d in log:
        d = {k: v.cpu().numpy() for k, v in d.items()}
        for k in range(K):
            s = o.step()
            assert np.array_equal(s["actions"], d["actions"][k]) and np.array_equal(s["obs"], d["obs"][k]), (gid, k)
            assert np.array_equal(s["reward"], d["reward"][k].astype(np.float64)) and np.array_equal(s["prob"], d["prob"][k].astype(np.float64))
            assert np.array_equal(s["terminated"], d["terminated"][k].astype(bool)) and np.array_equal(s["truncated"], d["truncated"][k].astype(bool))
            ended += int(s["terminated"].sum() + s["truncated"].sum())
    assert ended > 0
    r.close()


def test_blackjack_graph_replays_against_the_oracle():
    import torch
    from gym_amd.toy_text import BlackjackRollout
    from oracle.oracle import OracleBlackjack

    n, K, replays = 4096, 5, 6
    r = BlackjackRollout(n, seed=5, action_seed=6)
    obs0 = r.reset(seed=5).cpu().numpy().copy()
    out = r.trajectory_buffers(K, layout="separate", want_final=True)
    log = []
    with torch.cuda.stream(r.stream):
        r.handle.set_device_clock(True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=r.stream):
            r.rollout_per_step(K, out=out)
        for _ in range(replays):
            g.replay()
            log.append({k: v.clone() for k, v in out.items()})
    r.synchronize()
    assert r.handle.get_counters()[0] == replays * K
    o = OracleBlackjack(n, seed=5, action_seed=6, sab=True)
    assert np.array_equal(o.reset(seed=5), obs0)
    ended = 0
    for d in log:
        d = {k: v.cpu().numpy() for k, v in d.items()}
        for k in range(K):
            s = o.step()
            assert np.array_equal(s["actions"], d["actions"][k]) and np.array_equal(s["obs"], d["obs"][k]), k
            assert np.array_equal(s["reward"], d["reward"][k]) and np.array_equal(s["terminated"], d["terminated"][k].astype(bool))
            m = s["final_mask"]
            assert np.array_equal(s["final_obs"][:, m], d["final_obs"][k][:, m])
            ended += int(m.sum())
    assert ended > n
    r.close()
