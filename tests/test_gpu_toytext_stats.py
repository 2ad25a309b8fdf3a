"""gym.wrappers.RecordEpisodeStatistics (gym/wrappers/record_episode_statistics.py:96-151) over the toy_text engines on the MI355X
(VERDICT r5 item 3): the float32 return accumulator and the episode length fused into tab_step_kernel / tab_traj_kernel / bj_kernel.

  (a) the reference's own wrapper run (tests/golden/toytext_stats_*.npz) replayed with its recorded uniforms / cards through the C ABI:
      returns and lengths BIT FOR BIT, in both of the reference's dtypes;
  (b) the same goldens through the user-facing surface: gym_amd.RecordEpisodeStatistics(gym_amd.make(id, 8)) (infos["episode"]) and
      gym_amd.make(id, 8, wrappers=RecordEpisodeStatistics) (infos["final_info"][i]["episode"]), VectorListInfo on top;
  (c) Philox mode: the trajectory kernels' [K][N] statistics against the oracle twin with the oracle's restatement of the wrapper on top
      (pinned to the reference by tests/test_toytext_oracle.py), fused K-step launch == single steps; both dtype sets;
  (d) snapshots carry the running returns."""
import numpy as np
import pytest

from helpers import TOYTEXT_STATS_CASES, replay_toytext_stats, toytext_stats_start

pytestmark = pytest.mark.gpu


def _mdp(tag):
    from gym_amd import toy_text

    return toy_text.TOY_TEXT_REGISTRY[tag].build()


def _tab(mdp, n, limit, **kw):
    from gym_amd import _native

    return _native.Tab(mdp.num_states, mdp.num_actions, mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated,
                       mdp.initial_cum, n, limit, **kw)


class _Engine:
    """The C ABI directly: host steps with the recorded draws, statistics from *_episode_stats_host."""

    def __init__(self, g, n, tag):
        from gym_amd import _native

        self.g, self.tag = g, tag
        if tag == "Blackjack-v1":
            self.h = _native.Blackjack(n, natural=bool(g["natural"]), sab=bool(g["sab"]))
            self.h.episode_stats(True)
            self.h.reset_host(toytext_stats_start(g))
        else:
            mdp = _mdp(tag)
            self.h = _tab(mdp, n, int(g["max_episode_steps"]))
            self.h.episode_stats(True)
            self.h.reset_host()
            self.h.set_state(toytext_stats_start(g, mdp), np.zeros(n, np.int32))

    def step(self, t):
        g = self.g
        if self.tag == "Blackjack-v1":
            _, rew, term, trunc, _ = self.h.step_host(g["actions"][t], g["draws"][t].astype(np.int8))
        else:
            _, rew, term, trunc, _, _, _ = self.h.step_host(g["actions"][t], np.ascontiguousarray(g["draws"][t][:, :2].T))
        r, l = self.h.episode_stats_host()
        return rew, term, trunc, r, l


@pytest.mark.parametrize("tag", TOYTEXT_STATS_CASES)
def test_fused_accumulators_replay_the_reference_wrapper_bit_for_bit(tag):
    episodes, g = replay_toytext_stats(tag, lambda g, n: _Engine(g, n, tag))
    assert episodes == int(g["ep_mask"].sum())


def _inject(env, g, tag, clock):
    """The adapter steps with the reference's recorded draws: its handle's step_host gets them as the extra argument."""
    h = env.unwrapped.handle
    orig = h.step_host
    if tag == "Blackjack-v1":
        h.step_host = lambda a, cards=None, **kw: orig(a, g["draws"][clock[0]].astype(np.int8), **kw)
        h.reset_host(toytext_stats_start(g))
    else:
        h.step_host = lambda a, uniforms=None, **kw: orig(a, np.ascontiguousarray(g["draws"][clock[0]][:, :2].T), **kw)
        h.set_state(toytext_stats_start(g, _mdp(tag)), np.zeros(env.num_envs, np.int32))


@pytest.mark.parametrize("tag", TOYTEXT_STATS_CASES)
def test_wrappers_over_the_toy_text_adapters_report_what_the_reference_reports(tag):
    import gym_amd
    from gym_amd.wrappers import RecordEpisodeStatistics, VectorListInfo
    from helpers import GOLDEN
    import os

    g = np.load(os.path.join(GOLDEN, f"toytext_stats_{tag}.npz"))
    T, n = g["actions"].shape
    kw = dict(natural=bool(g["natural"]), sab=bool(g["sab"])) if tag == "Blackjack-v1" else {}
    # (1) the vector-level wrapper: infos["episode"] float64 arrays + "_episode"
    clock = [0]
    env = RecordEpisodeStatistics(gym_amd.make(tag, num_envs=n, **kw))
    assert env.episode_returns is None
    env.reset(seed=1)
    _inject(env, g, tag, clock)
    for t in range(T):
        clock[0] = t
        _, rew, term, trunc, infos = env.step(g["actions"][t])
        done = g["ep_mask"][t]
        assert np.array_equal(rew, g["reward"][t]) and np.array_equal(term | trunc, done)
        if done.any():
            ep = infos["episode"]
            assert ep["r"].dtype == np.float64 and ep["l"].dtype == np.float64 and np.array_equal(infos["_episode"], done)
            assert np.array_equal(ep["r"], g["ep_r"][t]) and np.array_equal(ep["l"], g["ep_l"][t]), t
        else:
            assert "episode" not in infos
    assert env.episode_count == int(g["ep_mask"].sum()) and env.episode_returns.dtype == np.float32
    env.close()
    # (2) the per-sub-env wrapper through make(wrappers=...), VectorListInfo on top
    clock = [0]
    env = gym_amd.make(tag, num_envs=n, wrappers=RecordEpisodeStatistics, **kw)
    assert type(env).__name__ == "SubEnvEpisodeStatistics"
    env = VectorListInfo(env)
    env.reset(seed=1)
    _inject(env, g, tag, clock)
    for t in range(T):
        clock[0] = t
        _, _, term, trunc, infos = env.step(g["actions"][t])
        assert isinstance(infos, list) and len(infos) == n
        for i in np.flatnonzero(g["ep_mask"][t]):
            ep = infos[i]["final_info"]["episode"]
            assert isinstance(ep["r"], np.float32) and isinstance(ep["l"], np.int32)
            assert ep["r"] == g["sub_ep_r"][t][i] and ep["l"] == g["sub_ep_l"][t][i], (t, i)
        for i in np.flatnonzero(~g["ep_mask"][t]):
            assert "final_info" not in infos[i]
    env.close()


@pytest.mark.parametrize("gid,limit,compact", [("FrozenLake-v1", 100, False), ("FrozenLake8x8-v1", 9, True), ("Taxi-v3", 13, False), ("Taxi-v3", 13, True)])
def test_tabular_trajectory_statistics_equal_the_oracle_and_single_steps(gid, limit, compact):
    """Philox mode, [K][N] statistics of the trajectory kernel (tab_traj_kernel: asserted) == K single-step launches of the general kernel
    == the oracle twin with the oracle's RecordEpisodeStatistics on top; running returns carried across launches."""
    import torch

    from gym_amd import _native
    from gym_amd.toy_text import TOY_TEXT_REGISTRY, TabularRollout
    from oracle.oracle import EpisodeStats, OracleTabEnv

    n, K = 3001, 48
    runs = {}
    for mode in ("fused", "single"):
        r = TabularRollout(gid, n, seed=31, action_seed=32, max_episode_steps=limit, compact=compact)
        r.handle.episode_stats(True)
        r.reset(seed=31)
        out = r.trajectory_buffers(2 * K)
        with torch.cuda.stream(r.stream):
            epr = torch.full((2 * K, n), -7.0, dtype=torch.float32, device=r.device)
            epl = torch.full((2 * K, n), -7, dtype=torch.int32, device=r.device)
        r.stream.synchronize()
        if mode == "fused":
            for half in (0, 1):          # two launches: the running returns cross the boundary
                sl = slice(half * K, (half + 1) * K)
                r.handle.set_episode_outputs(epr[sl], epl[sl])
                r.rollout_per_step(K, out={k: v[sl] for k, v in out.items()})
                assert r.handle.last_kernel() == _native.TAB_KERNEL_TRAJECTORY
        else:
            for k in range(2 * K):
                r.handle.set_episode_outputs(epr[k], epl[k])
                r.handle.rollout(1, out["obs"][k], out["reward"][k], out["terminated"][k], out["truncated"][k], out["prob"][k],
                                 actions_out_dev=out["actions"][k], per_step=False)
            assert r.handle.last_kernel() == _native.TAB_KERNEL_GENERAL
        r.synchronize()
        runs[mode] = ({k: v.cpu().numpy() for k, v in out.items()}, epr.cpu().numpy(), epl.cpu().numpy(),
                      r.handle.episode_stats_host(want_running=True)[2])
        r.close()
    a, b = runs["fused"], runs["single"]
    done = (a[0]["terminated"] | a[0]["truncated"]).astype(bool)
    assert done.sum() > 200 and np.array_equal(a[0]["reward"], b[0]["reward"])
    assert np.array_equal(a[1][done], b[1][done]) and np.array_equal(a[2][done], b[2][done]) and np.array_equal(a[3], b[3])
    assert np.isin(a[1][~done], (-7.0, 0.0)).all() and np.isin(a[2][~done], (-7, 0)).all()   # untouched, or the zeros of a wave that stored its whole row segment
    mdp = TOY_TEXT_REGISTRY[gid].build()
    orc = OracleTabEnv(mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated, mdp.initial_cum, n, limit, seed=31, action_seed=32)
    orc.reset()
    st = EpisodeStats(n)
    for k in range(2 * K):
        o = orc.step()
        er, el, m = st.step(o["reward"], o["terminated"], o["truncated"])
        assert np.array_equal(m, done[k]) and np.array_equal(a[1][k][m], er[m]) and np.array_equal(a[2][k][m], el[m]), k
    assert np.array_equal(a[3], st.returns)


@pytest.mark.parametrize("compact", [False, True])
def test_blackjack_trajectory_statistics_equal_the_oracle_and_single_steps(compact):
    import torch

    from gym_amd.toy_text import BlackjackRollout
    from oracle.oracle import EpisodeStats, OracleBlackjack

    n, K = 3001, 40
    runs = {}
    for mode in ("fused", "single"):
        r = BlackjackRollout(n, seed=3, action_seed=4, natural=True, sab=False, compact=compact)
        r.handle.episode_stats(True)
        r.reset(seed=3)
        out = r.trajectory_buffers(K, layout="separate")
        with torch.cuda.stream(r.stream):
            epr = torch.full((K, n), -7.0, dtype=torch.float32, device=r.device)
            epl = torch.full((K, n), -7, dtype=torch.int32, device=r.device)
        r.stream.synchronize()
        if mode == "fused":
            r.handle.set_episode_outputs(epr, epl)
            r.rollout_per_step(K, out=out)
        else:
            for k in range(K):
                r.handle.set_episode_outputs(epr[k], epl[k])
                r.handle.rollout(1, out["obs"][k], out["reward"][k], out["terminated"][k], out["truncated"][k], None, out["actions"][k], compact=compact)
        r.synchronize()
        runs[mode] = ({k: v.cpu().numpy() for k, v in out.items()}, epr.cpu().numpy(), epl.cpu().numpy(),
                      r.handle.episode_stats_host(want_running=True)[2], r.handle.snapshot())
        r.close()
    a, b = runs["fused"], runs["single"]
    done = (a[0]["terminated"] | a[0]["truncated"]).astype(bool)
    assert done.sum() > 1000 and np.array_equal(a[1][done], b[1][done]) and np.array_equal(a[2][done], b[2][done]) and np.array_equal(a[3], b[3])
    assert np.isin(a[1][~done], (-7.0, 0.0)).all()
    orc = OracleBlackjack(n, natural=True, sab=False, seed=3, action_seed=4)
    orc.reset()
    st = EpisodeStats(n)
    for k in range(K):
        o = orc.step()
        er, el, m = st.step(o["reward"], o["terminated"], o["truncated"])
        assert np.array_equal(m, done[k]) and np.array_equal(a[1][k][m], er[m]) and np.array_equal(a[2][k][m], el[m]), k
    assert set(np.unique(a[1][done])) <= {-1.0, 0.0, 1.0, 1.5}
    # (d) the snapshot carries the running returns and the draw contract; an old-contract snapshot is refused
    from gym_amd import _native

    snap = a[4]
    assert snap["stats_on"] and np.array_equal(snap["running_returns"], a[3]) and snap["draw_contract"] == _native.BJ_DRAW_CONTRACT
    h = _native.Blackjack(n, natural=True, sab=False)
    h.restore(snap)
    assert np.array_equal(h.episode_stats_host(want_running=True)[2], a[3])
    old = dict(snap, draw_contract=1)
    with pytest.raises(ValueError, match="draw contract"):
        h.restore(old)
    h.close()
