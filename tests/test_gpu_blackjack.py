"""Blackjack-v1 on the MI355X (mxv_bj_*, through the C ABI): the reference's games replayed with the reference's own cards
(exact), Philox mode against the oracle twin (exact; fused K-step launch == single steps == action tape), the
HipBlackjackVectorEnv surface, and properties at 2^20 tables."""
import numpy as np
import pytest

from helpers import BLACKJACK_CASES, replay_blackjack

pytestmark = pytest.mark.gpu


class _HipAdapter:
    def __init__(self, n, natural, sab):
        from gym_amd import _native

        self.h = _native.Blackjack(n, natural=natural, sab=sab)

    def reset(self, cards):
        return self.h.reset_host(cards)

    def step(self, actions, cards):
        obs, rew, term, trunc, fin = self.h.step_host(actions, cards)
        return dict(obs=obs, reward=rew, terminated=term, truncated=trunc, final_obs=fin)


@pytest.mark.parametrize("tag", BLACKJACK_CASES)
def test_device_replays_reference_games_exactly(tag):
    ndone, g = replay_blackjack(tag, _HipAdapter)
    assert ndone == int(g["final_mask"].sum()) > 1000


@pytest.mark.parametrize("rules", [dict(sab=True), dict(natural=True, sab=False)])
def test_philox_mode_equals_oracle_twin_fused_and_tape(rules):
    import torch
    from gym_amd import _native
    from oracle.oracle import OracleBlackjack

    n, K = 3001, 40
    dev = torch.device("cuda")
    runs = {}
    for mode in ("fused", "single", "tape"):
        h = _native.Blackjack(n, seed=3, action_seed=4, **rules)
        obs0 = h.reset_host()
        bufs = dict(obs=torch.zeros((K, 3, n), dtype=torch.int64, device=dev), actions=torch.zeros((K, n), dtype=torch.int64, device=dev),
                    reward=torch.zeros((K, n), dtype=torch.float64, device=dev), terminated=torch.zeros((K, n), dtype=torch.uint8, device=dev),
                    truncated=torch.zeros((K, n), dtype=torch.uint8, device=dev), final_obs=torch.zeros((K, 3, n), dtype=torch.int64, device=dev))
        torch.cuda.synchronize()
        if mode == "fused":
            h.rollout(K, bufs["obs"], bufs["reward"], bufs["terminated"], bufs["truncated"], bufs["final_obs"], bufs["actions"], per_step=True)
        elif mode == "single":
            for k in range(K):
                h.rollout(1, bufs["obs"][k], bufs["reward"][k], bufs["terminated"][k], bufs["truncated"][k], bufs["final_obs"][k], bufs["actions"][k])
        else:
            tape = torch.from_numpy(runs["fused"][1]["actions"]).to(dev)
            h.rollout(K, bufs["obs"], bufs["reward"], bufs["terminated"], bufs["truncated"], bufs["final_obs"], None, actions_tape_dev=tape, per_step=True)
            bufs["actions"] = tape
        h.sync()
        runs[mode] = (obs0, {k: v.cpu().numpy() for k, v in bufs.items()}, h.get_state())
        h.close()
    for mode in ("single", "tape"):
        assert np.array_equal(runs["fused"][0], runs[mode][0])
        for k, v in runs["fused"][1].items():
            assert np.array_equal(v, runs[mode][1][k]), (mode, k)
        assert np.array_equal(runs["fused"][2][0], runs[mode][2][0])
    orc = OracleBlackjack(n, seed=3, action_seed=4, natural=rules.get("natural", False), sab=rules.get("sab", False))
    assert np.array_equal(orc.reset(seed=3), runs["fused"][0])
    dev_out = runs["fused"][1]
    ndone = 0
    for k in range(K):
        o = orc.step()
        assert np.array_equal(o["actions"], dev_out["actions"][k]) and np.array_equal(o["obs"], dev_out["obs"][k]), k
        assert np.array_equal(o["reward"], dev_out["reward"][k]) and np.array_equal(o["terminated"], dev_out["terminated"][k].astype(bool))
        m = o["final_mask"]
        assert np.array_equal(o["final_obs"][:, m], dev_out["final_obs"][k][:, m])
        ndone += int(m.sum())
    assert ndone > n


def test_hip_blackjack_vector_env_contract():
    import gym_amd
    from gym_amd.spaces import MultiDiscrete, Tuple
    from gym_amd.toy_text import HipBlackjackVectorEnv

    env = gym_amd.make("Blackjack-v1", 16)
    assert isinstance(env, HipBlackjackVectorEnv) and env.sab and not env.natural
    assert isinstance(env.observation_space, Tuple) and isinstance(env.observation_space[0], MultiDiscrete)
    obs, infos = env.reset(seed=11)
    assert isinstance(obs, tuple) and len(obs) == 3 and all(o.dtype == np.int64 and o.shape == (16,) for o in obs)
    assert infos == {} and np.all((obs[0] >= 4) & (obs[0] <= 21)) and np.all((obs[1] >= 1) & (obs[1] <= 10))
    obs2, _ = env.reset(seed=11)
    assert all(np.array_equal(a, b) for a, b in zip(obs, obs2))
    env.action_space.seed(0)
    seen_final = False
    for _ in range(30):
        obs, rew, term, trunc, infos = env.step(env.action_space.sample())
        assert rew.dtype == np.float64 and set(np.unique(rew)) <= {-1.0, 0.0, 1.0} and not trunc.any()
        assert np.all(rew[~term] == 0.0)
        if term.any():
            fo = infos["final_observation"]
            assert fo.dtype == object and np.array_equal(infos["_final_observation"], term)
            i = int(np.flatnonzero(term)[0])
            assert isinstance(fo[i], tuple) and isinstance(fo[i][2], bool) and infos["final_info"][i] == {}
            seen_final = True
    assert seen_final
    with pytest.raises(AssertionError):
        env.step(np.full(16, 2))
    env.close()
    nat = gym_amd.make("Blackjack-v1", 8, natural=True, sab=False)
    assert nat.natural and not nat.sab
    nat.close()


def test_pickle_round_trip_continues_identically():
    import pickle

    import gym_amd

    env = gym_amd.make("Blackjack-v1", num_envs=64, natural=True, sab=False)
    env.reset(seed=5)
    env.action_space.seed(6)
    for _ in range(5):
        env.step(env.action_space.sample())
    twin = pickle.loads(pickle.dumps(env))
    assert twin.natural and not twin.sab
    for step in range(30):
        if step == 11:
            for u, v in zip(env.reset()[0], twin.reset()[0]):
                assert np.array_equal(u, v)
        a = env.action_space.sample()
        r0, r1 = env.step(a), twin.step(a)
        for u, v in zip(r0[0], r1[0]):
            assert np.array_equal(u, v)
        for x, y in zip(r0[1:4], r1[1:4]):
            assert np.array_equal(x, y)
    env.close()
    twin.close()


def test_full_size_properties():
    """2^20 tables, 32 sampled steps in one launch: sharding invariance and game statistics."""
    import torch
    from gym_amd import _native

    n, K = 1 << 20, 32
    dev = torch.device("cuda")

    def run(num, offset):
        h = _native.Blackjack(num, sab=True, seed=1, action_seed=2, env_offset=offset)
        o0 = torch.zeros((3, num), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()      # the handle launches on its own stream: allocations / fills must have landed
        h.reset(o0)
        obs = torch.zeros((K, 3, num), dtype=torch.int64, device=dev)
        rew = torch.zeros((K, num), dtype=torch.float64, device=dev)
        term = torch.zeros((K, num), dtype=torch.uint8, device=dev)
        act = torch.zeros((K, num), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        h.rollout(K, obs, rew, term, None, None, act, per_step=True)
        h.sync()
        h.close()
        return o0, obs, rew, term, act

    full = run(n, 0)
    parts = [run(n // 4, w * (n // 4)) for w in range(4)]
    for i in range(5):
        assert torch.equal(full[i], torch.cat([p[i] for p in parts], dim=-1)), i
    o0, obs, rew, term, act = full
    first = o0[1].cpu().numpy()
    frac10 = (first == 10).mean()
    assert abs(frac10 - 4 / 13) < 0.002                                  # the dealer shows a ten-valued card 4 times in 13
    r = rew.cpu().numpy()
    te = term.cpu().numpy().astype(bool)
    assert set(np.unique(r)) <= {-1.0, 0.0, 1.0} and np.all(r[~te] == 0.0)
    assert -0.5 < r[te].mean() < -0.3                                    # a random policy loses about 0.4 per game
    a = act.cpu().numpy()
    assert np.all(te[a == 0])                                            # sticking always ends the game
