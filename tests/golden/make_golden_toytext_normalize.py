#!/usr/bin/env python3
"""Golden vectors for the Normalize* wrappers over the toy_text engines, made by RUNNING THE REFERENCE's gym.wrappers.NormalizeReward /
NormalizeObservation (gym/wrappers/normalize.py:50-145) in the build container over the SAME seeded trajectories as
make_golden_toytext_stats.py (asserted: actions, flags and every recorded draw equal the committed toytext_stats_<tag>.npz, which holds the
driving data; this file adds only what the wrappers returned):

    python tests/golden/make_golden_toytext_normalize.py      -> tests/golden/toytext_normalize_<tag>.npz

  vec_reward   NormalizeReward(gym.vector.make(id, 8), gamma=0.97)                        the vector-level wrapper (batch statistics)
  sub_reward   gym.vector.make(id, 8, wrappers=partial(NormalizeReward, gamma=0.97))       one wrapper per sub-env (batches of one)
  vec_obs, vec_obs0   NormalizeObservation(gym.vector.make(id, 8)): float64 [N] (the Discrete observations of the tabular ids; Blackjack's
               Tuple observations have no shape and the reference's wrapper cannot be built over them); raw_reset1 / raw_reset2 = the raw
               states of the two reset() calls the wrapper folded into obs_rms before the first step"""
import functools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_toytext_stats as base  # noqa: E402  (numpy aliases, reference on the path, Recorder)

import gym  # noqa: E402
from gym.wrappers import NormalizeObservation, NormalizeReward  # noqa: E402

N = base.N


def run(gid, kw, T, how):
    if how == "sub_reward":
        venv = gym.vector.make(gid, num_envs=N, asynchronous=False, wrappers=functools.partial(NormalizeReward, gamma=0.97), **kw)
        top = venv
    else:
        venv = gym.vector.make(gid, num_envs=N, asynchronous=False, **kw)
        top = NormalizeReward(venv, gamma=0.97) if how == "vec_reward" else NormalizeObservation(venv)
    raws = [e.unwrapped for e in venv.envs]
    venv.action_space.seed(31)
    top.reset(seed=777)
    raw_reset1 = np.array([int(r.s) for r in raws], np.int64) if hasattr(raws[0], "s") else None      # what the wrapper's first update saw
    for r in raws:
        r._np_random = base.Recorder(r._np_random)
    obs0, _ = top.reset()
    raw_reset2 = np.array([int(r.s) for r in raws], np.int64) if hasattr(raws[0], "s") else None
    rec = {k: [] for k in ("actions", "obs", "reward", "terminated", "truncated")}
    for t in range(T):
        a = venv.action_space.sample()
        o, rw, te, tr, info = top.step(a)
        for k, v in (("actions", a), ("obs", np.asarray(o)), ("reward", np.asarray(rw)), ("terminated", np.asarray(te, dtype=bool)), ("truncated", np.asarray(tr, dtype=bool))):
            rec[k].append(v)
    out = {k: np.stack(v) for k, v in rec.items()}
    out["obs0"] = np.asarray(obs0)
    if raw_reset1 is not None:
        out["raw_reset1"], out["raw_reset2"] = raw_reset1, raw_reset2
    return out


def main():
    for tag, (gid, kw, T) in base.CASES.items():
        g = np.load(os.path.join(HERE, f"toytext_stats_{tag}.npz"))
        out = {}
        for how in ("vec_reward", "sub_reward") + (() if tag == "Blackjack-v1" else ("vec_obs",)):
            r = run(gid, kw, T, how)
            for k in ("actions", "terminated", "truncated"):
                assert np.array_equal(r[k], g[k]), (tag, how, k)        # the committed trajectory, wrapped another way
            if how == "vec_obs":
                assert r["obs"].dtype == np.float64 and r["obs"].shape == (T, N) and np.array_equal(r["reward"], g["reward"])
                out["vec_obs"], out["vec_obs0"], out["raw_reset1"], out["raw_reset2"] = r["obs"], r["obs0"], r["raw_reset1"], r["raw_reset2"]
            else:
                assert r["reward"].dtype == np.float64
                out[how] = r["reward"]
        out["gamma"] = np.float64(0.97)
        path = os.path.join(HERE, f"toytext_normalize_{tag}.npz")
        np.savez_compressed(path, **out)
        print(tag, {k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if k != "gamma"}, os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
