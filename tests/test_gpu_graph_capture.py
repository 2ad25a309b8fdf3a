"""A caller's hipGraph around the engine (mxv_set_device_clock, DeviceRollout.enable_graph_capture): the learner-in-the-loop path
(SyncVectorEnv.step with a policy between the steps, gym/vector/sync_vector_env.py:131-169) recorded ONCE with torch.cuda.graph —
policy kernels and env steps together — and replayed.  A replayed graph must continue every Philox stream where single calls would:
the step index travels in device memory, not as a capture-time kernel argument.  Held against the same calls made one by one, bit for
bit, for the env kinds whose step reads the index (sampled actions: all; step(actions): Acrobot with torque noise) and for CartPole
with a policy in the loop."""
import numpy as np
import pytest

from helpers import ENV_IDS, GYM_IDS, LIMITS

pytestmark = pytest.mark.gpu


def _policy(torch, obs, W):
    return (obs @ W).argmax(dim=1)


def _run(kind, n, per_graph, replays, captured, noise=0.0, sampled=False):
    import torch
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout(GYM_IDS[kind], n, seed=3, action_seed=4, max_episode_steps=min(LIMITS[kind], 60))
    if noise:
        p = r.handle.get_params()
        p[10] = noise
        r.handle.set_params(p)
    r.reset(seed=3)
    dev = r.device
    torch.manual_seed(0)
    W = torch.randn(r.O, max(r.NA, 2), device=dev)
    steps = per_graph * replays
    log = {k: [] for k in ("obs", "reward", "terminated", "truncated", "actions")}

    def one():
        if sampled:
            r.step_sampled(record_actions=True)
            acts = r.actions
        else:
            acts = _policy(torch, r.obs, W).to(r.action_dtype)
            r.step(acts)
        return acts

    with torch.cuda.stream(r.stream):
        if captured:
            r.enable_graph_capture()
            # static logs: every step of the graph writes its own row
            bufs = {"obs": torch.empty((per_graph, n, r.O), device=dev), "reward": torch.empty((per_graph, n), dtype=torch.float64, device=dev),
                    "terminated": torch.empty((per_graph, n), dtype=torch.uint8, device=dev), "truncated": torch.empty((per_graph, n), dtype=torch.uint8, device=dev),
                    "actions": torch.empty((per_graph, n), dtype=r.action_dtype, device=dev)}
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=r.stream):
                for k in range(per_graph):
                    a = one()
                    bufs["obs"][k].copy_(r.obs); bufs["reward"][k].copy_(r.reward); bufs["terminated"][k].copy_(r.terminated)
                    bufs["truncated"][k].copy_(r.truncated); bufs["actions"][k].copy_(a.reshape(n))
            for _ in range(replays):
                g.replay()
                for k in log:
                    log[k].append(bufs[k].clone())
            out = {k: torch.cat(v) for k, v in log.items()}
        else:
            for _ in range(steps):
                a = one()
                for k, v in (("obs", r.obs), ("reward", r.reward), ("terminated", r.terminated), ("truncated", r.truncated), ("actions", a.reshape(n))):
                    log[k].append(v.clone())
            out = {k: torch.stack(v) for k, v in log.items()}
    r.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res["state"] = r.handle.get_state()
    res["t"] = r.handle.get_counters()[0]
    res["episodes"] = r.handle.get_episodes()
    r.close()
    return res


@pytest.mark.parametrize("kind,noise,sampled", [("CartPole", 0.0, False), ("Acrobot", 0.4, False), ("CartPole", 0.0, True), ("Pendulum", 0.0, True),
                                                ("MountainCar", 0.0, True)])
def test_replayed_graph_equals_single_calls(kind, noise, sampled):
    n, per_graph, replays = 4096, 8, 12
    a = _run(kind, n, per_graph, replays, captured=False, noise=noise, sampled=sampled)
    b = _run(kind, n, per_graph, replays, captured=True, noise=noise, sampled=sampled)
    assert a["t"] == b["t"] == per_graph * replays          # the device clock, read back through mxv_get_counters
    for k in ("obs", "reward", "terminated", "truncated", "actions"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (kind, k)
    assert np.array_equal(a["state"][0], b["state"][0]) and np.array_equal(a["state"][1], b["state"][1])
    assert np.array_equal(a["episodes"], b["episodes"])
    assert int((a["terminated"] | a["truncated"]).sum()) > 0
    if sampled:
        assert len(np.unique(a["actions"][:8], axis=0)) > 1     # the sampled actions change from step to step (the clock runs)


def test_device_clock_round_trips_and_fused_rollouts_follow_it():
    """Clock on -> calls -> off: the host index picks up where the device left it; fused, eager and internal-graph rollouts and a
    seed() in between behave as without the clock."""
    import torch
    from gym_amd.rollout import DeviceRollout

    runs = []
    for clock in (False, True):
        r = DeviceRollout("CartPole-v1", 2048, seed=9, action_seed=10)
        r.reset(seed=9)
        if clock:
            r.enable_graph_capture()
        seq = []
        for mode in ("fused", "eager", "graph", "fused"):
            r.rollout(5, mode=mode, record_actions=True)
            r.synchronize()
            seq.append((r.obs.cpu().numpy().copy(), r.actions.cpu().numpy().copy()))
        assert r.handle.get_counters()[0] == 20
        r.handle.set_counters(100, 0)
        r.step_sampled()
        r.synchronize()
        seq.append((r.obs.cpu().numpy().copy(), r.actions.cpu().numpy().copy()))
        assert r.handle.get_counters()[0] == 101
        if clock:
            r.enable_graph_capture(False)
            assert r.handle.get_counters()[0] == 101
        r.step_sampled()
        r.synchronize()
        seq.append((r.obs.cpu().numpy().copy(), r.actions.cpu().numpy().copy()))
        r.handle.seed(9)
        assert r.handle.get_counters()[0] == 0
        runs.append(seq)
        r.close()
    for (o1, a1), (o2, a2) in zip(*runs):
        assert np.array_equal(o1, o2) and np.array_equal(a1, a2)


def test_graphed_loop_helper_records_a_trajectory():
    """DeviceRollout.graphed_loop: policy + step recorded once; the on_step hook fills the caller's static [K, N] tensors; three
    replays == 3 K single calls (the warm-up steps of the helper included in both)."""
    import torch
    from gym_amd.rollout import DeviceRollout

    n, K = 2048, 16
    torch.manual_seed(1)
    W = torch.randn(4, 2, device="cuda")
    policy = lambda obs: (obs @ W).argmax(dim=1)   # noqa: E731

    a = DeviceRollout("CartPole-v1", n, seed=11, action_seed=12, max_episode_steps=20)   # (this policy balances: let TimeLimit end episodes)
    a.reset(seed=11)
    traj = {"obs": torch.empty((K, n, 4), device="cuda"), "done": torch.empty((K, n), dtype=torch.uint8, device="cuda")}

    def record(k):
        traj["obs"][k].copy_(a.obs)
        torch.bitwise_or(a.terminated, a.truncated, out=traj["done"][k])

    g = a.graphed_loop(policy, K, warmup=3, on_step=record)
    got = []
    for _ in range(3):
        g.replay()
        a.synchronize()
        got.append((traj["obs"].cpu().numpy().copy(), traj["done"].cpu().numpy().copy()))
    assert a.handle.get_counters()[0] == 3 + 3 * K

    b = DeviceRollout("CartPole-v1", n, seed=11, action_seed=12, max_episode_steps=20)
    b.reset(seed=11)
    with torch.cuda.stream(b.stream):
        for _ in range(3):
            b.step(policy(b.obs))
        for rep in range(3):
            for k in range(K):
                b.step(policy(b.obs))
                b.stream.synchronize()
                assert np.array_equal(got[rep][0][k], b.obs.cpu().numpy()), (rep, k)
                assert np.array_equal(got[rep][1][k], (b.terminated | b.truncated).cpu().numpy()), (rep, k)
    assert sum(int(d.sum()) for _, d in got) > 0
    a.close(), b.close()


@pytest.mark.parametrize("gid,compact", [("FrozenLake8x8-v1", False), ("Taxi-v3", True)])
def test_tabular_rollouts_in_a_callers_graph(gid, compact):
    """mxv_tab_set_device_clock: K-step sampled rollouts of the table engine (trajectory kernel) and single general-kernel steps recorded
    in a torch.cuda.graph and replayed == the same calls one by one."""
    import torch
    from gym_amd.toy_text import TabularRollout

    n, K, replays = 8192, 6, 7          # K = 6: launches start and end inside the four-step action blocks, at a different phase every replay
    runs = []
    for captured in (False, True):
        r = TabularRollout(gid, n, seed=3, action_seed=4, max_episode_steps=9, compact=compact)
        r.reset(seed=3)
        out = r.trajectory_buffers(K, layout="separate")
        log = []
        with torch.cuda.stream(r.stream):
            if captured:
                r.handle.set_device_clock(True)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=r.stream):
                    r.rollout_per_step(K, out=out)
                    r.handle.rollout(1, out["obs"][0], out["reward"][0], out["terminated"][0], out["truncated"][0], out["prob"][0],
                                     actions_out_dev=out["actions"][0], per_step=False)          # the general kernel, one step
                for _ in range(replays):
                    g.replay()
                    log.append({k: v.clone() for k, v in out.items()})
            else:
                for _ in range(replays):
                    r.rollout_per_step(K, out=out)
                    r.handle.rollout(1, out["obs"][0], out["reward"][0], out["terminated"][0], out["truncated"][0], out["prob"][0],
                                     actions_out_dev=out["actions"][0], per_step=False)
                    log.append({k: v.clone() for k, v in out.items()})
        r.synchronize()
        assert r.handle.get_counters()[0] == replays * (K + 1)
        runs.append(([{k: v.cpu().numpy() for k, v in d.items()} for d in log], r.handle.get_state()))
        r.close()
    for a, b in zip(runs[0][0], runs[1][0]):
        for k in a:
            assert np.array_equal(a[k], b[k]), (gid, k)
    assert all(np.array_equal(x, y) for x, y in zip(runs[0][1], runs[1][1]))
    assert len({tuple(d["actions"][1][:64]) for d in runs[1][0]}) > 1        # the action stream moves from replay to replay


def test_blackjack_rollouts_in_a_callers_graph():
    import torch
    from gym_amd import _native

    n, K, replays = 4096, 5, 6
    dev = torch.device("cuda", 0)
    runs = []
    for captured in (False, True):
        h = _native.Blackjack(n, seed=5, action_seed=6)
        st = torch.cuda.Stream()
        h.set_stream(st.cuda_stream)
        obs = torch.empty((K, 3, n), dtype=torch.int64, device=dev)
        rew = torch.empty((K, n), dtype=torch.float64, device=dev)
        term, trunc = (torch.empty((K, n), dtype=torch.uint8, device=dev) for _ in range(2))
        act = torch.empty((K, n), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        h.reset()
        log = []
        with torch.cuda.stream(st):
            if captured:
                h.set_device_clock(True)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    h.rollout(K, obs, rew, term, trunc, None, actions_out_dev=act, per_step=True)
                for _ in range(replays):
                    g.replay()
                    log.append([x.clone() for x in (obs, rew, term, act)])
            else:
                for _ in range(replays):
                    h.rollout(K, obs, rew, term, trunc, None, actions_out_dev=act, per_step=True)
                    log.append([x.clone() for x in (obs, rew, term, act)])
        h.sync()
        assert h.get_counters()[0] == replays * K
        runs.append(([[x.cpu().numpy() for x in row] for row in log], h.get_state()))
        h.close()
    for a, b in zip(runs[0][0], runs[1][0]):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert all(np.array_equal(x, y) for x, y in zip(runs[0][1], runs[1][1]))


def test_the_graphed_policy_gradient_example_learns():
    """examples/policy_gradient_graphed.py: REINFORCE with the 64-step sampling loop (policy, env steps, trajectory copies, fused
    episode statistics) replayed from one hipGraph; the policy improves, and the recorded loop is faster than one call per step."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("policy_gradient_graphed", os.path.join(root, "examples", "policy_gradient_graphed.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    h = mod.train(iterations=14, verbose=False)
    assert h[0]["mean_episode_length"] < 30 and h[-1]["mean_episode_length"] > 45, [round(x["mean_episode_length"], 1) for x in h]
    e = mod.train(iterations=4, graphed=False, verbose=False)
    fast = min(x["us_per_step"] for x in h[2:])
    assert fast < 0.8 * min(x["us_per_step"] for x in e[1:]), (fast, [x["us_per_step"] for x in e])


def test_recording_without_the_device_clock_is_refused():
    """a step / rollout call on a stream that is being captured, with the step index still travelling by value, would replay
    the same draws for ever; the C ABI returns MXV_ERR_UNSUPPORTED instead (and works again once the capture has ended)."""
    import torch
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout("CartPole-v1", 2048, seed=1, action_seed=2)
    r.reset(seed=1)
    out = r.trajectory_buffers(4, layout="separate")
    r.rollout_per_step(4, out=out)
    r.synchronize()
    codes = []
    with torch.cuda.stream(r.stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=r.stream):
            for call in (lambda: r.handle.step_sampled(r.obs, r.reward, r.terminated, r.truncated, None, r.actions),
                         lambda: r.handle.rollout(4, out["obs"], out["reward"], out["terminated"], out["truncated"], None, out["actions"],
                                                  per_step=True, mode=_native.ROLLOUT_FUSED)):
                try:
                    call()
                    codes.append(None)
                except _native.MxvError as e:
                    codes.append(e.code)
            r.obs.add_(0.0)              # (an empty capture is not worth instantiating)
    assert codes == [_native.ERR_UNSUPPORTED, _native.ERR_UNSUPPORTED]
    t0 = r.handle.get_counters()[0]
    r.rollout_per_step(4, out=out)        # outside a capture: as before
    r.synchronize()
    assert r.handle.get_counters()[0] == t0 + 4 == 8
    r.close()
