#!/usr/bin/env python3
"""Golden vectors for Blackjack-v1 (gym/envs/toy_text/blackjack.py) made by RUNNING THE REFERENCE (build container only):

    python tests/golden/make_golden_blackjack.py          -> tests/golden/blackjack_<tag>.npz

gym.vector.make("Blackjack-v1", num_envs=8, asynchronous=False, **kwargs) with seeded random actions; every card the sub-envs
draw (np_random.choice(deck), blackjack.py:18) is recorded in order, so the engine and the oracle can be dealt the same cards
and must reproduce observations (a tuple of three int64 arrays), rewards, flags and final_observation exactly."""
import os
import sys

import numpy as np

for _name, _val in (("bool8", np.bool_), ("float_", np.float64)):
    if not hasattr(np, _name):
        setattr(np, _name, _val)
sys.path.insert(0, "/root/reference")
import warnings  # noqa: E402

import gym  # noqa: E402

warnings.filterwarnings("ignore")
gym.logger.set_level(gym.logger.ERROR)
HERE = os.path.dirname(os.path.abspath(__file__))
N, T, MAX_DRAWS = 8, 300, 24
CASES = {"sab": {}, "natural": {"natural": True, "sab": False}, "plain": {"natural": False, "sab": False}}


class Recorder:
    def __init__(self, g):
        self.g, self.cards = g, []

    def choice(self, a, *args, **kw):
        v = self.g.choice(a, *args, **kw)
        if isinstance(a, list) and len(a) == 13:
            self.cards.append(int(v))
        return v

    def __getattr__(self, name):
        return getattr(self.g, name)


def main():
    for tag, kw in CASES.items():
        venv = gym.vector.make("Blackjack-v1", num_envs=N, asynchronous=False, **kw)
        raws = [e.unwrapped for e in venv.envs]
        venv.action_space.seed(5)
        venv.reset(seed=100)                      # creates the generators; re-deal below with recorded cards
        for r in raws:
            r._np_random = Recorder(r._np_random)
        obs0, _ = venv.reset()
        cards0 = np.array([r._np_random.cards[:4] for r in raws], np.int8)
        assert isinstance(obs0, tuple) and len(obs0) == 3 and all(o.dtype == np.int64 for o in obs0)
        acts = np.zeros((T, N), np.int64); cards = np.zeros((T, N, MAX_DRAWS), np.int8); ncards = np.zeros((T, N), np.int32)
        obs = np.zeros((T, 3, N), np.int64); rew = np.zeros((T, N)); term = np.zeros((T, N), np.bool_); trunc = np.zeros((T, N), np.bool_)
        fin = np.zeros((T, 3, N), np.int64); fmask = np.zeros((T, N), np.bool_)
        for t in range(T):
            a = venv.action_space.sample()
            for r in raws:
                r._np_random.cards.clear()
            o, rw, te, tr, info = venv.step(a)
            acts[t], rew[t], term[t], trunc[t] = a, rw, te, tr
            obs[t] = np.stack(o)
            for i, r in enumerate(raws):
                c = r._np_random.cards
                assert len(c) <= MAX_DRAWS
                cards[t, i, :len(c)] = c
                ncards[t, i] = len(c)
            if "final_observation" in info:
                fmask[t] = info["_final_observation"]
                assert info["final_observation"].dtype == object
                for i in np.flatnonzero(fmask[t]):
                    fo = info["final_observation"][i]
                    assert isinstance(fo, tuple) and info["final_info"][i] == {}
                    fin[t, :, i] = [int(fo[0]), int(fo[1]), int(fo[2])]
            assert np.array_equal(fmask[t], te | tr)
        out = os.path.join(HERE, f"blackjack_{tag}.npz")
        np.savez_compressed(out, natural=np.bool_(kw.get("natural", False)), sab=np.bool_(kw.get("sab", True)), cards0=cards0,
                            obs0=np.stack(obs0), actions=acts, cards=cards, ncards=ncards, obs=obs, reward=rew, terminated=term,
                            truncated=trunc, final_obs=fin, final_mask=fmask)
        print(f"{tag}: T={T} done={int(fmask.sum())} max draws/step={int(ncards.max())} rewards={sorted(set(rew.ravel()))} -> "
              f"{os.path.getsize(out)} B")


if __name__ == "__main__":
    main()
