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


def _packed(p_sum, p_ace, p_two, dfirst, d_sum, d_ace, d_two):
    """One table's state word (gym_amd/csrc/mxv_bj.hip: player sum | ace | two cards | dealer's first card | dealer sum | ace | two cards)."""
    return p_sum | (p_ace << 6) | (p_two << 7) | (dfirst << 8) | (d_sum << 12) | (d_ace << 18) | (d_two << 19)


def test_device_cards_are_deck_draws_in_every_role():
    """The round-5 draw contract measured ON THE DEVICE, role by role, at 2^20 tables (not through the twin): from a hand that cannot bust
    the hit card is the change of the player's raw sum (card 0); sticking on a hard 19 against a dealer that starts from a known hand is
    won / drawn / lost with the probabilities of iid deck draws (cards 0..3 and the later calls: a dealer starting at 2 draws five cards
    and more in several percent of the games); the hands dealt after those games (cards 4..7) show the dealer card, and the player
    totals, of two deck cards each."""
    import torch
    from gym_amd import _native
    from helpers import DECK_P, chi2_ok, dealer_score_distribution

    n = 1 << 20
    dev = torch.device("cuda")
    h = _native.Blackjack(n, sab=False, seed=77, action_seed=78)
    h.reset_host()
    obs = torch.zeros((3, n), dtype=torch.int64, device=dev)
    rew = torch.zeros(n, dtype=torch.float64, device=dev)
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    hit, stick = torch.ones(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    # card 0 as the hit card: player raw sum 2 without an ace (cannot bust), three cards on the table already
    h.set_state(np.full(n, _packed(2, 0, 0, 10, 20, 0, 1), np.int32), np.zeros(n, np.int32), t=5, r=1)
    h.step(hit, obs, rew, term)
    h.sync()
    st, _ = h.get_state()
    assert not term.any().item() and not rew.any().item()
    card = (st & 63) - 2
    ok, chi2 = chi2_ok(np.bincount(card, minlength=11)[1:], DECK_P)
    assert ok and card.min() == 1 and card.max() == 10, chi2
    # the dealer's draws: exact win / draw / loss probabilities of a hard 19
    for d_sum, d_ace in ((2, 0), (12, 0), (6, 1), (16, 0)):
        h.set_state(np.full(n, _packed(19, 0, 0, 2, d_sum, d_ace, 1), np.int32), np.zeros(n, np.int32), t=100 + d_sum, r=1)
        h.step(stick, obs, rew, term)
        h.sync()
        assert term.all().item()
        dist = dealer_score_distribution(d_sum, bool(d_ace))
        p_win, p_draw = sum(p for s, p in dist.items() if s < 19), dist.get(19, 0.0)
        r = rew.cpu().numpy()
        ok, chi2 = chi2_ok([(r == 1.0).sum(), (r == 0.0).sum(), (r == -1.0).sum()], [p_win, p_draw, 1.0 - p_win - p_draw])
        assert ok, (d_sum, d_ace, chi2)
        # ... and every table was dealt new hands from cards 4..7 of the same call
        o = obs.cpu().numpy()
        ok, chi2 = chi2_ok(np.bincount(o[1], minlength=11)[1:], DECK_P)                      # the dealer's first card = card 4
        assert ok, ("dealer shows", d_sum, chi2)
        totals = np.zeros(22)
        for c1 in range(1, 11):
            for c2 in range(1, 11):
                s_, a_ = c1 + c2, c1 == 1 or c2 == 1
                totals[s_ + 10 if a_ and s_ + 10 <= 21 else s_] += DECK_P[c1 - 1] * DECK_P[c2 - 1]
        ok, chi2 = chi2_ok(np.bincount(o[0], minlength=22)[4:22], totals[4:22])             # the player's total = cards 6, 7
        assert ok, ("player total", d_sum, chi2)
        st, _ = h.get_state()
        dsum = np.zeros(21)
        for c1 in range(1, 11):
            for c2 in range(1, 11):
                dsum[c1 + c2] += DECK_P[c1 - 1] * DECK_P[c2 - 1]
        ok, chi2 = chi2_ok(np.bincount((st >> 12) & 63, minlength=21)[2:21], dsum[2:21])    # the dealer's raw sum = cards 4 + 5
        assert ok, ("dealer sum", d_sum, chi2)
    h.close()


@pytest.mark.parametrize("tape", [False, True])
def test_compact_outputs_equal_the_reference_dtypes(tape):
    """mxv_bj_rollout_compact (int32 observations / actions, float32 rewards: the contract's 4-byte scalars) == mxv_bj_rollout value for
    value, sampled actions and an action tape, natural rule (1.5 is exact in float32), ragged size."""
    import torch
    from gym_amd import _native

    n, K = 70_001, 48
    dev = torch.device("cuda")
    outs = []
    for compact in (False, True):
        it, ft = (torch.int32, torch.float32) if compact else (torch.int64, torch.float64)
        h = _native.Blackjack(n, natural=True, sab=False, seed=8, action_seed=9, max_episode_steps=3)
        h.reset_host()
        b = dict(obs=torch.zeros((K, 3, n), dtype=it, device=dev), reward=torch.zeros((K, n), dtype=ft, device=dev),
                 terminated=torch.zeros((K, n), dtype=torch.uint8, device=dev), truncated=torch.zeros((K, n), dtype=torch.uint8, device=dev),
                 final_obs=torch.zeros((K, 3, n), dtype=it, device=dev), actions=torch.zeros((K, n), dtype=it, device=dev))
        tape_dev = None
        if tape:
            g = torch.Generator().manual_seed(3)
            tape_dev = torch.randint(0, 2, (K, n), generator=g, dtype=torch.int64).to(dev)
        torch.cuda.synchronize()
        h.rollout(K, b["obs"], b["reward"], b["terminated"], b["truncated"], b["final_obs"], None if tape else b["actions"],
                  actions_tape_dev=tape_dev, per_step=True, compact=compact)
        h.sync()
        outs.append(({k: v.cpu().numpy() for k, v in b.items()}, h.get_state()))
        h.close()
    (a, sa), (c, sc) = outs
    for k in a:
        assert c[k].dtype.itemsize <= 4 and np.array_equal(a[k], c[k].astype(a[k].dtype)), k
    assert np.array_equal(sa[0], sc[0]) and np.array_equal(sa[1], sc[1])
    assert 1.5 in a["reward"] and a["truncated"].any() and a["terminated"].any()
