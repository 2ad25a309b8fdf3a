"""The integer form of categorical_sample the tabular trajectory kernel uses (gym_amd/csrc/mxv_tab.hip: tab_traj_kernel compares the
Philox word against thresholds packed at create time) is THE comparison of the reference, `np.cumsum(prob_n) > np_random.random()`
(gym/envs/toy_text/utils.py:4-8), for every 32-bit word: checked here on the host function the packing uses (mxv_tab_word_threshold),
at the boundaries where a rounding in either form would show.  No device needed."""
import numpy as np

from gym_amd import _native


def _u(w):
    return (np.asarray(w, np.float64) + 0.5) * 2.0 ** -32        # the engine's uniform for word w (exact in fp64)


def _thr(c):
    return int(_native.lib.mxv_tab_word_threshold(float(c)))


def test_threshold_is_the_float_comparison_for_every_word():
    rng = np.random.default_rng(0)
    cums = [0.0, -1.0, 1.0, 1.0 - 2.0 ** -53, 1.0 + 2.0 ** -52, 1.0 / 3.0, 2.0 / 3.0, 1.0 / 3.0 + 1.0 / 3.0, 0.1, 0.8, 2.0 ** -33,
            2.0 ** -33 + 2.0 ** -80, 2.0 ** -32, 1.5 * 2.0 ** -32, 1 - 2.0 ** -33, 1 - 2.0 ** -33 + 2.0 ** -53, 5e-324, 1e-300]
    for k in rng.integers(0, 2 ** 32, 200):                    # cumulative probabilities exactly ON a word's uniform, and one ulp to each side
        c = float(_u(int(k)))
        cums += [c, np.nextafter(c, 0.0), np.nextafter(c, 2.0), float(k) * 2.0 ** -32, np.nextafter(float(k) * 2.0 ** -32, 2.0)]
    cums += list(rng.random(300)) + list(np.cumsum(rng.dirichlet(np.ones(5), 60), axis=1).ravel())
    for c in cums:
        T = _thr(c)
        assert 0 <= T <= 2 ** 32
        near = {0, 1, 2 ** 32 - 1, 2 ** 32 - 2, 2 ** 31} | {min(max(T + d, 0), 2 ** 32 - 1) for d in (-2, -1, 0, 1, 2)}
        w = np.array(sorted(near) + list(rng.integers(0, 2 ** 32, 64)), dtype=np.uint64)
        assert np.array_equal(c > _u(w), w < T), (c, T)


def test_the_registered_envs_pack():
    """What pack_fast_table requires of an MDP, restated on the host tables: every list's cumulative probabilities end in a value no
    word reaches (T == 2^32), no list starts with a threshold of 0, every reward is a float32 value — true for FrozenLake (slippery or
    not), Taxi and CliffWalking, so their trajectory launches take the specialised kernel (the GPU suite asserts that they do)."""
    from gym_amd.toy_text import TOY_TEXT_REGISTRY

    for gid, spec in TOY_TEXT_REGISTRY.items():
        mdp = spec.build()
        M = mdp.max_transitions
        assert M in (1, 3), gid
        cum = np.asarray(mdp.cum_prob).reshape(-1, M)
        for row in cum:
            valid = row[row >= 0]
            assert len(valid) >= 1 and np.all(np.diff(valid) >= 0)
            assert _thr(valid[-1]) == 2 ** 32 and all(_thr(c) >= 1 for c in valid)
        rew = np.asarray(mdp.reward, np.float64)
        assert np.array_equal(rew.astype(np.float32).astype(np.float64), rew)
        ic = np.asarray(mdp.initial_cum)
        assert np.all(np.diff(ic) >= 0) and _thr(ic[-1]) == 2 ** 32
