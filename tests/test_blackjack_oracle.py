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
