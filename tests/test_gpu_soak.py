"""Parity at length, in the driver-run suite (Next #1b / #1c).

* The kernel bench.py times — rollout_kernel_v3<CartPole, default attributes, two envs per lane, unguarded trigonometry, trajectory
  outputs float64 + int64> at 2^20 envs, 256 steps per launch — compared DIRECTLY with the oracle: observations, rewards, both masks and
  the sampled actions of the first and the last 4096 envs of the batch, every step of 768, the instantiation asserted through
  mxv_last_launch.  (Before: the bench line's work_check covered flags and actions; observations and rewards of this instantiation met
  the oracle only through "fused == single-step launches bit for bit".)
* A bounded soak, all five env kinds: fused K-step launches == one launch per step == tape-driven launches, bit for bit, on every
  output of every step, over 1.3e8 env-steps; and the fused kernel against the oracle over 1.6e7 env-steps (short launches with the
  fp64 state handed to the oracle between them, so that last-bit libm differences cannot grow in the chaotic kinds)."""
import numpy as np
import pytest

from helpers import (DISCRETE, ENV_IDS, ENV_NAMES, GYM_IDS, LIMITS, MAX_OBS_ULPS, REWARD_ATOL, REWARD_ATOL_DEFAULT, REWARD_RTOL,
                     OracleEngine, ulps32)

pytestmark = pytest.mark.gpu


def _oracle_window(name, n_total, lo, count, seed, action_seed):
    """Oracle twin of envs [lo, lo + count) of a logical vector env (global indices: the RNG contract is sharding-invariant)."""
    o = OracleEngine(name, count, LIMITS[name], seed=seed, action_seed=action_seed, env_offset=lo).o
    o.reset(seed=seed)
    return o


def test_the_timed_instantiation_against_the_oracle():
    import torch
    from gym_amd.rollout import DeviceRollout

    n, K, W, chunks = 1 << 20, 256, 4096, 3
    r = DeviceRollout("CartPole-v1", n, seed=0, action_seed=1)          # bench.py's seeds
    obs0 = r.reset(seed=0)
    wins = [(0, _oracle_window("CartPole", n, 0, W, 0, 1)), (n - W, _oracle_window("CartPole", n, n - W, W, 0, 1))]
    traj = r.trajectory_buffers(K, layout="separate")
    ended = 0
    for c in range(chunks):
        out = r.rollout_per_step(K, out=traj)
        r.synchronize()
        info = r.handle.last_launch()
        assert info == dict(info, kernel=1, env_id=0, param_mode=1, envs_per_lane=2, safe=0, out_mode=1, tape=0, steps=K, grid=n // 128,
                            block=64), info
        for lo, o in wins:
            d = {k: out[k][:, lo:lo + W].cpu().numpy() for k in ("obs", "reward", "terminated", "truncated", "actions")}
            for k in range(K):
                a = o.sample_actions()
                ro, rr, rte, rtr, _, _ = o.step(a)
                tag = f"chunk {c} step {k} envs [{lo}, {lo + W})"
                assert np.array_equal(d["actions"][k], a), tag
                assert np.array_equal(d["terminated"][k].astype(bool), rte) and np.array_equal(d["truncated"][k].astype(bool), rtr), tag
                assert np.array_equal(d["reward"][k], rr), tag
                u = ulps32(d["obs"][k], ro)
                assert u.max() <= MAX_OBS_ULPS, f"{tag}: observations {u.max()} float32 ulps apart"
                ended += int((rte | rtr).sum())
    assert ended > 0.03 * chunks * K * 2 * W        # autoresets inside the compared windows (~1 per 22 env-steps)
    st, el = r.handle.get_state()
    for lo, o in wins:                               # 768 free-running steps later: fp64 state and counters
        np.testing.assert_allclose(st[:, lo:lo + W], o.state, rtol=1e-9, atol=1e-12)
        assert np.array_equal(el[lo:lo + W], o.elapsed)
    r.close()


@pytest.mark.parametrize("name", ENV_NAMES)
def test_bounded_soak_fused_tape_per_step_and_oracle(name):
    import torch
    from gym_amd.rollout import DeviceRollout

    # (1) fused == per-step == tape, bit for bit: 65 536 envs x 3 x 128 steps x 3 engines
    n, K, chunks = 1 << 16, 128, 3
    kw = dict(seed=1234, action_seed=4321, env_offset=5 << 20, max_episode_steps=min(LIMITS[name], 150))   # episodes end in every kind
    eng = [DeviceRollout(GYM_IDS[name], n, **kw) for _ in range(3)]
    for e in eng:
        e.reset(seed=1234)
    done = 0
    for c in range(chunks):
        fa = eng[0].rollout_per_step(K, mode="fused")
        fb = eng[1].rollout_per_step(K, mode="eager")
        fc = eng[2].rollout_tape(fa["actions"].clone())
        for e in eng:
            e.synchronize()
        assert eng[0].handle.last_launch()["kernel"] == 1 and eng[1].handle.last_launch()["kernel"] == 0
        assert eng[2].handle.last_launch()["tape"] == 1 and eng[2].handle.last_launch()["kernel"] == 1
        for key in ("obs", "reward", "terminated", "truncated"):
            assert torch.equal(fa[key], fb[key]), (name, c, key, "fused vs per-step")
            assert torch.equal(fa[key], fc[key]), (name, c, key, "fused vs tape")
        assert torch.equal(fa["actions"], fb["actions"])
        done += int((fa["terminated"] | fa["truncated"]).sum())
    ref_state = eng[0].handle.get_state()
    for other in eng[1:]:
        for x, y in zip(ref_state, other.handle.get_state()):
            assert np.array_equal(x, y, equal_nan=False)
        assert np.array_equal(eng[0].handle.get_episodes(), other.handle.get_episodes())
    for e in eng:
        e.close()
    assert done > 0
    compared = 3 * n * K * chunks

    # (2) the fused kernel against the oracle: 16 384 envs, 48 launches of 8 steps, trajectories compared step by step, the device's
    # fp64 state copied into the oracle between launches
    n, K, launches = 1 << 14, 8, 48
    r = DeviceRollout(GYM_IDS[name], n, seed=77, action_seed=78, max_episode_steps=min(LIMITS[name], 120))
    o = OracleEngine(name, n, min(LIMITS[name], 120), seed=77, action_seed=78).o
    assert ulps32(r.reset(seed=77).cpu().numpy(), o.reset(seed=77)).max() <= MAX_OBS_ULPS
    ndone = 0
    for i in range(launches):
        st, el = r.handle.get_state()
        assert np.array_equal(el, o.elapsed) and np.array_equal(r.handle.get_episodes(), o.episodes)
        o.state[:] = st
        out = r.rollout_per_step(K, mode="fused")
        r.synchronize()
        assert r.handle.last_launch()["kernel"] == 1
        d = {k: v.cpu().numpy() for k, v in out.items() if k in ("obs", "reward", "terminated", "truncated", "actions")}
        for k in range(K):
            a = o.sample_actions()
            ro, rr, rte, rtr, _, _ = o.step(a)
            tag = f"{name} launch {i} step {k}"
            assert np.array_equal(d["actions"][k].reshape(-1), a), tag
            assert np.array_equal(d["terminated"][k].astype(bool), rte) and np.array_equal(d["truncated"][k].astype(bool), rtr), tag
            assert ulps32(d["obs"][k], ro).max() <= MAX_OBS_ULPS, tag
            np.testing.assert_allclose(d["reward"][k], rr, rtol=REWARD_RTOL, atol=REWARD_ATOL.get(name, REWARD_ATOL_DEFAULT), err_msg=tag)
            ndone += int((rte | rtr).sum())
    r.close()
    assert ndone > n            # every env finished at least one episode (TimeLimit 120 at most)
    print(f"{name}: {compared:.2e} env-steps fused == per-step == tape, {n * K * launches:.2e} fused vs oracle, {ndone} episodes ended")


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("name,n,epl", [("CartPole", 1 << 20, 2), ("Pendulum", 1 << 19, 1), ("Acrobot", 1 << 19, 1), ("MountainCar", 1 << 19, 2),
                                        ("MountainCarContinuous", 1 << 19, 2), ("CartPole", 1 << 17, 1)])
def test_every_benchmarked_instantiation_equals_per_step_launches(name, n, epl, compact):
    """The kernels behind bench.py's variants — every env kind at its BASELINE.json size, both dtype sets (OUT = 1: the reference's
    float64 rewards / int64 actions, OUT = 2: float32 / int32), plus the 2^17-env share of an 8-GPU job — are the instantiations they
    are supposed to be (mxv_last_launch) and equal one launch per step (step_kernel, itself held against the oracle at these sizes by
    tests/test_gpu_parity.py) bit for bit on every output of every step."""
    import torch
    from gym_amd.rollout import DeviceRollout

    K = 48
    kw = dict(seed=21, action_seed=22, reward_f32=compact, action_i32=compact, max_episode_steps=min(LIMITS[name], 40))
    a, b = DeviceRollout(GYM_IDS[name], n, **kw), DeviceRollout(GYM_IDS[name], n, **kw)
    a.reset(seed=21), b.reset(seed=21)
    ta, tb = a.trajectory_buffers(K, layout="separate"), b.trajectory_buffers(K, layout="separate")
    for rep in range(2):
        fa = a.rollout_per_step(K, mode="fused", out=ta)
        fb = b.rollout_per_step(K, mode="eager", out=tb)
        a.synchronize(), b.synchronize()
        li = a.handle.last_launch()
        assert (li["kernel"], li["envs_per_lane"], li["safe"], li["out_mode"], li["tape"], li["steps"]) == (1, epl, 0, 2 if compact else 1, 0, K), li
        assert b.handle.last_launch()["kernel"] == 0
        for key in ("obs", "reward", "terminated", "truncated", "actions"):
            assert fa[key].dtype == fb[key].dtype and torch.equal(fa[key], fb[key]), (name, n, compact, rep, key)
        assert int((fa["terminated"] | fa["truncated"]).sum()) > 0
    for x, y in zip(a.handle.get_state(), b.handle.get_state()):
        assert np.array_equal(x, y)
    a.close(), b.close()
