"""Blackjack-v1 on the CPU: the oracle (oracle/tabular.c, card-list hands like the reference) dealt the cards the reference
drew reproduces the reference's SyncVectorEnv trajectories exactly (all three rule sets), and its Philox draw stream is
invariant under sharding."""
import numpy as np
import pytest

from helpers import BLACKJACK_CASES, replay_blackjack
from oracle.oracle import OracleBlackjack


class _Adapter:
    def __init__(self, n, natural, sab):
        self.o = OracleBlackjack(n, natural=natural, sab=sab)

    def reset(self, cards):
        return self.o.reset(cards=cards)

    def step(self, actions, cards):
        return self.o.step(actions, cards)


@pytest.mark.parametrize("tag", BLACKJACK_CASES)
def test_oracle_replays_reference_games_exactly(tag):
    ndone, g = replay_blackjack(tag, _Adapter)
    assert ndone == int(g["final_mask"].sum()) > 1000
    if tag == "natural":
        assert 1.5 in g["reward"]


def test_philox_draws_are_shard_invariant_and_look_like_a_deck():
    n, T = 256, 60
    full = OracleBlackjack(n, seed=9, action_seed=10)
    halves = [OracleBlackjack(n // 2, seed=9, action_seed=10, env_offset=o) for o in (0, n // 2)]
    assert np.array_equal(full.reset(seed=9), np.concatenate([h.reset(seed=9) for h in halves], axis=1))
    showing = []
    for _ in range(T):
        a = full.step()
        b = [h.step() for h in halves]
        for k in ("actions", "reward", "terminated"):
            assert np.array_equal(a[k], np.concatenate([x[k] for x in b])), k
        assert np.array_equal(a["obs"], np.concatenate([x["obs"] for x in b], axis=1))
        showing.append(a["obs"][1])
    counts = np.bincount(np.concatenate(showing), minlength=11)[1:]
    assert counts[:9].min() > 0 and 2.5 < counts[9] / counts[:9].mean() < 6.0   # four ten-valued ranks out of thirteen
    with pytest.raises(AssertionError):
        full.step(np.full(n, 2))


def test_two_cards_per_word_are_deck_draws_position_by_position_and_pairwise():
    """The round-5 card map (oracle/tabular.c bj_stream_card = gym_amd/csrc/mxv_bj.hip cards_of): two base-13 digits per Philox word.
    Every one of the first 16 card positions of a step's stream is a deck draw (chi-square against [1/13] * 9 + [4/13]), the two cards of a
    word are independent of each other, and so are neighbouring words' cards (chi-square of the 10 x 10 tables against the products)."""
    from helpers import DECK_P, chi2_ok
    from oracle.oracle import _p, lib

    n, count = 400_000, 16
    cards = np.zeros((n, count), np.int8)
    lib().orc_bj_stream_cards(n, 12345, 7, count, _p(cards))
    assert cards.min() == 1 and cards.max() == 10
    for g in range(count):
        ok, chi2 = chi2_ok(np.bincount(cards[:, g], minlength=11)[1:], DECK_P)
        assert ok, (g, chi2)
    joint_p = np.outer(DECK_P, DECK_P)
    for g0, g1 in ((0, 1), (2, 3), (6, 7), (14, 15), (1, 2), (3, 4), (7, 8), (0, 4)):      # same word, neighbouring words, call 0 / call 1
        table = np.zeros((10, 10))
        np.add.at(table, (cards[:, g0] - 1, cards[:, g1] - 1), 1)
        ok, chi2 = chi2_ok(table, joint_p)
        assert ok, (g0, g1, chi2)
    # another step, another env range: different cards (the counter carries t, the key the env's seed)
    other = np.zeros((n, count), np.int8)
    lib().orc_bj_stream_cards(n, 12345, 8, count, _p(other))
    assert 0.6 < (other != cards).mean() < 0.95


def test_oracle_games_have_the_exact_statistics_of_iid_deck_draws():
    """Twin games in Philox mode: sticking on a hard 19 against a dealer that starts from a known hand wins / draws / loses with the
    probabilities the dealer recursion gives for iid deck draws (helpers.dealer_score_distribution) — the fixed card roles (cards 0..3
    dealer draws, 4..7 the next hands, later draws from the following calls) do not bias a game."""
    from helpers import chi2_ok, dealer_score_distribution

    n = 200_000
    for dealer_sum, dealer_ace in ((4, False), (12, False), (6, True), (16, False)):
        o = OracleBlackjack(n, sab=False, seed=21 + dealer_sum, action_seed=1)
        o.reset(seed=21 + dealer_sum)
        o.player[:] = 0
        o.player[:, :2] = (10, 9)
        o.player[:, 32] = 2                                            # hard 19, not a natural
        o.dealer[:] = 0
        o.dealer[:, :2] = (1, dealer_sum - 1) if dealer_ace else ((dealer_sum + 1) // 2, dealer_sum // 2)
        o.dealer[:, 32] = 2
        out = o.step(np.zeros(n, np.int64))
        dist = dealer_score_distribution(dealer_sum, dealer_ace)
        p_win = sum(p for s, p in dist.items() if s < 19)
        p_draw = dist.get(19, 0.0)
        counts = [(out["reward"] == r).sum() for r in (1.0, 0.0, -1.0)]
        ok, chi2 = chi2_ok(counts, [p_win, p_draw, 1.0 - p_win - p_draw])
        assert ok and out["terminated"].all(), (dealer_sum, dealer_ace, counts, chi2)
