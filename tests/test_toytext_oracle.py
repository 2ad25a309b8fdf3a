"""SURVEY.md §8(f)-4 on the CPU: (1) the product's MDP builders (gym_amd/toy_text.py, host code) against the reference's
own `env.P` tables dumped into the goldens; (2) the oracle (oracle/tabular.c) replaying the reference's SyncVectorEnv
trajectories with the reference's recorded uniforms: BIT-EXACT observations, rewards, flags, infos; (3) the Philox
contract's shard invariance on the oracle."""
import sys

import numpy as np
import pytest

from helpers import TOYTEXT_CASES, load_toytext_golden, replay_toytext, toytext_mdp
from oracle.oracle import OracleTabEnv

REF_MAP_SEED3_SIZE6 = ['SFFFHH', 'FFFFFF', 'FFFFFF', 'FFFFFF', 'FFFFFH', 'HFHHFG']


@pytest.mark.parametrize("tag", TOYTEXT_CASES)
def test_mdp_tables_equal_reference(tag):
    g = load_toytext_golden(tag)
    mdp = toytext_mdp(g)
    cnt = g["table_num_transitions"]
    S, A = cnt.shape
    assert (mdp.num_states, mdp.num_actions, mdp.max_transitions) == (S, A, g["table_prob"].shape[2])
    live = np.arange(mdp.max_transitions)[None, None, :] < cnt[:, :, None]
    assert np.array_equal(mdp.cum_prob >= 0, live)
    for name, arr in (("prob", mdp.prob), ("next_state", mdp.next_state), ("reward", mdp.reward), ("terminated", mdp.terminated)):
        assert np.array_equal(arr[live], g[f"table_{name}"][live]), name
    # cum_prob is np.cumsum of each list (toy_text/utils.py:6-7)
    for s in range(S):
        for a in range(A):
            k = cnt[s, a]
            assert np.array_equal(mdp.cum_prob[s, a, :k], np.cumsum(g["table_prob"][s, a, :k]))
    assert np.array_equal(mdp.initial_distrib, g["table_initial_distrib"])
    assert mdp.reset_prob_is_int == bool(g["reset_prob_dtype_is_int"])
    if "table_action_mask" in g:
        assert np.array_equal(mdp.action_mask, g["table_action_mask"])
    else:
        assert mdp.action_mask is None


class _OracleAdapter:
    def __init__(self, mdp, n, limit):
        self.o = OracleTabEnv(mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated, mdp.initial_cum, n, limit)

    def set_state(self, state, elapsed):
        self.o.state[:] = state
        self.o.elapsed[:] = elapsed

    def step(self, actions, uniforms):
        return self.o.step(actions, uniforms)


@pytest.mark.parametrize("tag", TOYTEXT_CASES)
def test_oracle_replays_reference_trajectories_bit_exact(tag):
    g = load_toytext_golden(tag)
    ndone = replay_toytext(g, _OracleAdapter)
    assert ndone == int(g["final_mask"].sum()) and ndone > 0


def test_oracle_philox_streams_are_shard_invariant():
    g = load_toytext_golden("FrozenLake8x8-v1")
    mdp = toytext_mdp(g)
    n, T = 64, 40
    args = (mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated, mdp.initial_cum)
    full = OracleTabEnv(*args, n, 200, seed=3, action_seed=4)
    halves = [OracleTabEnv(*args, n // 2, 200, seed=3, action_seed=4, env_offset=o) for o in (0, n // 2)]
    o_full = full.reset(seed=3)
    o_half = np.concatenate([h.reset(seed=3) for h in halves])
    assert np.array_equal(o_full, o_half)
    for _ in range(T):
        a = full.step()
        b = [h.step() for h in halves]
        for k in ("actions", "obs", "reward", "terminated", "truncated", "prob"):
            assert np.array_equal(a[k], np.concatenate([x[k] for x in b])), k
    assert full.state.max() > 0


def test_invalid_action_raises_keyerror_like_the_reference():
    g = load_toytext_golden("Taxi-v3")
    mdp = toytext_mdp(g)
    o = OracleTabEnv(mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated, mdp.initial_cum, 4, 200)
    o.reset(seed=0)
    with pytest.raises(KeyError):
        o.step(np.array([0, 6, 1, 2]))


def test_taxi_action_mask_restates_reference_test():
    """tests/envs/test_env_implementation.py:129-136: an action is masked in exactly the states it cannot change."""
    from gym_amd.toy_text import taxi_mdp

    mdp = taxi_mdp()
    for state in range(mdp.num_states):
        for action, possible in enumerate(mdp.action_mask[state]):
            _, next_state, _, _ = mdp.transitions(state, action)[0]
            assert (state != next_state) if possible else (state == next_state)


def test_taxi_encode_decode_restates_reference_test():
    """tests/envs/test_env_implementation.py:139-148 (encode(decode(s)) == s), here over all 500 states, and against the live
    reference's TaxiEnv when it is importable."""
    from gym_amd.toy_text import taxi_decode, taxi_encode

    for state in range(500):
        assert taxi_encode(*taxi_decode(state)) == state
    try:
        sys.path.insert(0, "/root/reference")
        from gym.envs.toy_text.taxi import TaxiEnv
    except Exception:
        return
    finally:
        if sys.path and sys.path[0] == "/root/reference":
            sys.path.pop(0)
    env = TaxiEnv()
    for state in range(0, 500, 7):
        assert list(env.decode(state)) == list(taxi_decode(state))
        assert env.encode(*env.decode(state)) == taxi_encode(*taxi_decode(state))


@pytest.mark.parametrize("map_size", [5, 10, 16])
def test_frozenlake_dfs_map_generation_restates_reference_test(map_size):
    """tests/envs/test_env_implementation.py:97-126: every generated map has a path from S to G."""
    from gym_amd.toy_text import frozen_lake_mdp, generate_random_map

    np.random.seed(map_size)
    new_frozenlake = generate_random_map(map_size)
    assert len(new_frozenlake) == map_size and len(new_frozenlake[0]) == map_size
    directions = [(1, 0), (0, 1), (-1, 0), (0, -1)]
    frontier, discovered = [(0, 0)], set()
    found = False
    while frontier and not found:
        row, col = frontier.pop()
        if (row, col) not in discovered:
            discovered.add((row, col))
            for dr, dc in directions:
                nr, nc = row + dr, col + dc
                if 0 <= nr < map_size and 0 <= nc < map_size:
                    if new_frozenlake[nr][nc] == "G":
                        found = True
                    if new_frozenlake[nr][nc] not in "#H":
                        frontier.append((nr, nc))
    assert found, "No path through the frozenlake was found."
    mdp = frozen_lake_mdp(desc=new_frozenlake)               # and it is a well-formed MDP for the engine
    assert mdp.num_states == map_size * map_size and mdp.initial_distrib[0] == 1.0


def test_generate_random_map_equals_reference_under_the_same_global_seed():
    """Same draws from NumPy's global generator -> the same board as gym.envs.toy_text.frozen_lake.generate_random_map
    (values recorded from the reference in this container: np.random.seed(3); generate_random_map(6))."""
    from gym_amd.toy_text import generate_random_map

    np.random.seed(3)
    assert generate_random_map(6) == REF_MAP_SEED3_SIZE6


# ---- episode statistics: the oracle engines + the oracle's RecordEpisodeStatistics restatement against the reference's own wrapper -------
@pytest.mark.parametrize("tag", ["FrozenLake-v1", "Taxi-v3", "Blackjack-v1"])
def test_oracle_episode_statistics_replay_the_reference_wrapper(tag):
    """tests/golden/toytext_stats_<tag>.npz: gym.wrappers.RecordEpisodeStatistics run by the reference over the toy_text vector envs, both
    ways it can be applied.  The oracle's tabular / Blackjack engines dealt the recorded draws, with oracle.EpisodeStats
    (record_episode_statistics.py:119-143 restated) on top, reproduce returns and lengths bit for bit — which pins the checker the
    device's fused accumulators are held against (tests/test_gpu_toytext_stats.py)."""
    from helpers import replay_toytext_stats, toytext_stats_start
    from oracle.oracle import EpisodeStats, OracleBlackjack, OracleTabEnv

    class Eng:
        def __init__(self, g, n):
            self.g, self.stats = g, EpisodeStats(n)
            if tag == "Blackjack-v1":
                self.e = OracleBlackjack(n, natural=bool(g["natural"]), sab=bool(g["sab"]))
                self.e.reset(cards=toytext_stats_start(g))
            else:
                from gym_amd import toy_text

                mdp = toy_text.TOY_TEXT_REGISTRY[tag].build()
                self.e = OracleTabEnv(mdp.cum_prob, mdp.prob, mdp.next_state, mdp.reward, mdp.terminated, mdp.initial_cum, n,
                                      int(g["max_episode_steps"]))
                self.e.reset()
                self.e.state[:] = toytext_stats_start(g, mdp)
                self.e.elapsed[:] = 0

        def step(self, t):
            g = self.g
            if tag == "Blackjack-v1":
                out = self.e.step(g["actions"][t], g["draws"][t].astype(np.int8))
            else:
                out = self.e.step(g["actions"][t], np.ascontiguousarray(g["draws"][t][:, :2].T))
            r, l, _ = self.stats.step(out["reward"], out["terminated"], out["truncated"])
            return out["reward"], out["terminated"].astype(bool), out["truncated"].astype(bool), r, l

    episodes, g = replay_toytext_stats(tag, Eng)
    assert episodes == int(g["ep_mask"].sum())
