"""SURVEY.md §8(f)-1 on the MI355X: RecordEpisodeStatistics fused into the step kernels, against (a) the reference's own
wrapper riding the golden SyncVectorEnv trajectories (teacher-forced replay), (b) the oracle twin on seeded rollouts,
(c) the fused K-step launch vs single-step launches; plus the wrapper surface restating
tests/wrappers/test_record_episode_statistics.py and tests/wrappers/test_vector_list_info.py."""
import numpy as np
import pytest

from helpers import ENV_IDS, ENV_NAMES, GYM_IDS, LIMITS, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["default", "short"])
@pytest.mark.parametrize("name", ENV_NAMES)
def test_golden_replay_with_reference_wrapper_statistics(name, tag):
    from gym_amd import _native

    g = load_golden(name, f"p2_{tag}")
    T, N = g["action"].shape
    h = _native.Handle(ENV_IDS[name], N, int(g["max_episode_steps"]))
    h.episode_stats(True)
    h.reset_host()
    n_ep = 0
    for t in range(T):
        h.set_state(g["state_pre"][t].T, g["elapsed_pre"][t])
        _, rew, term, trunc, _ = h.step_host(g["action"][t])
        r, l, running = h.episode_stats_host(want_running=True)
        done = term | trunc
        assert np.array_equal(done, g["ep_mask"][t].astype(bool))
        # rewards are within rtol 1e-13 of the reference's; the float32 accumulators agree to 1 ulp, lengths exactly
        np.testing.assert_allclose(r[done], g["ep_return"][t][done], rtol=3e-7, atol=0)
        assert np.array_equal(l[done], g["ep_length"][t][done])
        np.testing.assert_allclose(running, g["ep_running_return"][t], rtol=3e-7, atol=1e-30)
        n_ep += int(done.sum())
    assert n_ep == int(g["ep_mask"].sum())
    h.close()


@pytest.mark.parametrize("name", ENV_NAMES)
def test_fused_rollout_statistics_equal_single_steps_and_oracle_twin(name):
    import torch
    from gym_amd.rollout import DeviceRollout
    from oracle.oracle import EpisodeStats

    n, K, limit = 3000, 45, 11
    runs = {}
    for mode in ("fused", "eager", "graph"):
        r = DeviceRollout(GYM_IDS[name], n, seed=8, action_seed=9, max_episode_steps=limit)
        r.enable_episode_stats()
        r.reset(seed=8)
        out = r.rollout_per_step(K, mode=mode)
        out2 = r.rollout_per_step(K, mode=mode, out=out if False else None)  # second chunk: accumulators carry over
        r.synchronize()
        runs[mode] = [{k: v.cpu().numpy() for k, v in o.items()} for o in (out, out2)]
        r.close()
    for mode in ("eager", "graph"):
        for a, b in zip(runs["fused"], runs[mode]):
            done = (a["terminated"] | a["truncated"]).astype(bool)
            assert np.array_equal(a["ep_return"][done], b["ep_return"][done]), mode
            assert np.array_equal(a["ep_length"][done], b["ep_length"][done]), mode
    st = EpisodeStats(n)
    episodes = 0
    for chunk in runs["fused"]:
        for k in range(K):
            er, elen, m = st.step(chunk["reward"][k], chunk["terminated"][k], chunk["truncated"][k])
            assert np.array_equal(er[m].view(np.uint32), chunk["ep_return"][k][m].view(np.uint32)), (name, k)
            assert np.array_equal(elen[m], chunk["ep_length"][k][m])
            assert np.all(chunk["ep_length"][k][m] <= limit)
            episodes += int(m.sum())
    assert episodes >= n * (2 * K // limit)


def test_wrapped_env_pickles_with_its_statistics():
    """RecordEpisodeStatistics around the engine pickles as one object: queues, counts and the running returns inside the
    engine continue identically in the copy."""
    import pickle

    import gym_amd

    env = gym_amd.RecordEpisodeStatistics(gym_amd.make("CartPole-v1", num_envs=32, max_episode_steps=12), deque_size=50)
    env.reset(seed=2)
    env.action_space.seed(3)
    for _ in range(20):
        env.step(env.action_space.sample())
    twin = pickle.loads(pickle.dumps(env))
    assert twin.episode_count == env.episode_count and list(twin.return_queue) == list(env.return_queue)
    assert np.array_equal(twin.episode_returns, env.episode_returns)
    for _ in range(40):
        a = env.action_space.sample()
        r0, r1 = env.step(a), twin.step(a)
        for x, y in zip(r0[:4], r1[:4]):
            assert np.array_equal(x, y)
        assert ("episode" in r0[4]) == ("episode" in r1[4])
        if "episode" in r0[4]:
            assert np.array_equal(r0[4]["episode"]["r"], r1[4]["episode"]["r"])
            assert np.array_equal(r0[4]["episode"]["l"], r1[4]["episode"]["l"])
    assert twin.episode_count == env.episode_count and list(twin.length_queue) == list(env.length_queue)
    env.close()
    twin.close()


def test_final_obs_and_statistics_recorded_together():
    """Trajectory tensors with final_obs AND the fused statistics in one launch: both equal what separate runs record."""
    import torch

    from gym_amd.rollout import DeviceRollout

    K, n = 48, 2500
    runs = {}
    for tag, stats, final in (("both", True, True), ("stats", True, False), ("final", False, True)):
        r = DeviceRollout("CartPole-v1", n, seed=8, action_seed=9)
        if stats:
            r.enable_episode_stats()
        r.reset(seed=8)
        out = r.trajectory_buffers(K, want_final=final)
        assert ("final_obs" in out) == final and ("ep_return" in out) == stats
        r.rollout_per_step(K, out=out)
        r.synchronize()
        runs[tag] = {k: v.clone() for k, v in out.items()}
        r.close()
    both, done = runs["both"], (runs["both"]["terminated"] | runs["both"]["truncated"]).bool()
    assert done.any()
    for k in ("obs", "reward", "terminated", "truncated", "actions"):
        assert torch.equal(both[k], runs["stats"][k]) and torch.equal(both[k], runs["final"][k]), k
    assert torch.equal(both["final_obs"][done], runs["final"]["final_obs"][done])
    assert torch.equal(both["ep_return"][done], runs["stats"]["ep_return"][done])
    assert torch.equal(both["ep_length"][done], runs["stats"]["ep_length"][done])


@pytest.mark.parametrize("env_id", ["CartPole-v1", "Pendulum-v1"])
@pytest.mark.parametrize("deque_size", [2, 5])
def test_wrapper_like_the_reference_test(env_id, deque_size):
    import gym_amd

    env = gym_amd.RecordEpisodeStatistics(gym_amd.make(env_id, num_envs=1, max_episode_steps=40), deque_size)
    assert env.episode_returns is None and env.episode_lengths is None
    for n in range(5):
        env.reset()
        assert env.episode_returns is not None and env.episode_lengths is not None
        assert env.episode_returns[0] == 0.0 and env.episode_lengths[0] == 0
        for t in range(40):
            _, _, terminated, truncated, info = env.step(env.action_space.sample())
            if terminated[0] or truncated[0]:
                assert "episode" in info and all(item in info["episode"] for item in ["r", "l", "t"])
                break
    assert len(env.return_queue) == deque_size and len(env.length_queue) == deque_size
    env.close()


@pytest.mark.parametrize("num_envs", [1, 4, 256])
def test_wrapper_with_vector_env_layout(num_envs):
    import gym_amd

    envs = gym_amd.RecordEpisodeStatistics(gym_amd.make("CartPole-v1", num_envs=num_envs))
    envs.reset(seed=3)
    envs.action_space.seed(3)
    saw = False
    ret = np.zeros(num_envs, dtype=np.float32)
    length = np.zeros(num_envs, dtype=np.int64)
    for _ in range(501):
        _, rew, terminateds, truncateds, infos = envs.step(envs.action_space.sample())
        ret += rew
        length += 1
        if any(terminateds) or any(truncateds):
            saw = True
            assert "episode" in infos and "_episode" in infos
            assert all(infos["_episode"] == np.bitwise_or(terminateds, truncateds))
            assert all(item in infos["episode"] for item in ["r", "l", "t"])
            m = infos["_episode"]
            assert infos["episode"]["r"].dtype == np.float64 and infos["episode"]["r"].shape == (num_envs,)
            assert np.array_equal(infos["episode"]["r"][m], ret[m]) and np.all(infos["episode"]["r"][~m] == 0)
            assert np.array_equal(infos["episode"]["l"][m], length[m]) and np.all(infos["episode"]["t"][m] > 0)
            assert "final_observation" in infos  # the engine's own infos are still there
            ret[m] = 0
            length[m] = 0
        else:
            assert "episode" not in infos and "_episode" not in infos
    assert saw and envs.episode_count == len(envs.return_queue) or envs.episode_count >= 100
    assert np.array_equal(envs.episode_returns, ret) and np.array_equal(envs.episode_lengths, length)
    envs.close()


def test_vector_list_info_and_wrong_wrapping_order():
    import gym_amd

    env = gym_amd.VectorListInfo(gym_amd.RecordEpisodeStatistics(gym_amd.make("CartPole-v1", num_envs=5)))
    _, info = env.reset(seed=1)
    assert isinstance(info, list) and len(info) == 5
    env.action_space.seed(1)
    for _ in range(80):
        _, _, term, trunc, list_info = env.step(env.action_space.sample())
        assert isinstance(list_info, list) and len(list_info) == 5
        for i in range(5):
            if term[i] or trunc[i]:
                assert {"episode", "final_observation", "final_info"} <= set(list_info[i])
                assert set(list_info[i]["episode"]) == {"r", "l", "t"}
            else:
                assert list_info[i] == {}
    env.close()
    wrong = gym_amd.VectorListInfo(gym_amd.make("CartPole-v1", num_envs=3))
    with pytest.raises(TypeError):  # the reference asserts at step time (test_wrong_wrapping_order); here at construction
        gym_amd.RecordEpisodeStatistics(wrong)
    wrong.close()


@pytest.mark.parametrize("deque_size", [100, 7])
def test_large_env_statistics_travel_packed_and_equal_the_dense_path(deque_size):
    """Above 2 MiB of step I/O RecordEpisodeStatistics reads (env index, return, length) of the finished envs from the packed
    record of the step (mxv_final_packed_stats_view) and builds infos["episode"] lazily.  Same wrapper output as the dense
    path (mxv_episode_stats_host + np.where over N, what small envs use), step for step: arrays, mask, queues, count —
    also through VectorListInfo."""
    import gym_amd

    n, limit = 120_000, 9
    env = gym_amd.RecordEpisodeStatistics(gym_amd.make("CartPole-v1", num_envs=n, max_episode_steps=limit), deque_size=deque_size)
    assert env.unwrapped._packed
    env.reset(seed=21)
    env.action_space.seed(22)
    want_r, want_l, count = [], [], 0
    for t in range(30):
        obs, rew, term, trunc, infos = env.step(env.action_space.sample())
        done = term | trunc
        if not done.any():
            assert "episode" not in infos
            continue
        r, l = env.unwrapped.handle.episode_stats_host()        # dense [N] staging of the same step
        ep = infos["episode"]
        assert np.array_equal(infos["_episode"], done) and infos["_episode"] is not done
        assert ep["r"].dtype == ep["l"].dtype == ep["t"].dtype == np.float64 and ep["r"].shape == (n,)
        assert np.array_equal(ep["r"], np.where(done, r, 0).astype(np.float64))
        assert np.array_equal(ep["l"], np.where(done, l, 0).astype(np.float64))
        assert np.array_equal(ep["t"] > 0, done)
        idx = np.flatnonzero(done)
        want_r += r[idx].tolist()
        want_l += l[idx].tolist()
        count += idx.size
        assert env.episode_count == count
        assert list(env.return_queue) == want_r[-deque_size:] and list(env.length_queue) == want_l[-deque_size:]
    assert count > 2 * n
    env.close()
    # VectorListInfo on top: the per-env dicts carry the same numbers
    env = gym_amd.VectorListInfo(gym_amd.RecordEpisodeStatistics(gym_amd.make("CartPole-v1", num_envs=n, max_episode_steps=3)))
    env.reset(seed=1)
    env.action_space.seed(2)
    for t in range(3):
        obs, rew, term, trunc, infos = env.step(env.action_space.sample())
    r, l = env.unwrapped.handle.episode_stats_host()
    done = term | trunc
    assert done.all() and isinstance(infos, list) and len(infos) == n
    for i in (0, 1, n // 2, n - 1):
        assert infos[i]["episode"]["r"] == float(r[i]) and infos[i]["episode"]["l"] == float(l[i]) == 3.0
        assert infos[i]["final_observation"].shape == (4,)
    env.close()
