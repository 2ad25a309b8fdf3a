#!/usr/bin/env python3
"""Golden vectors for the episode-statistics layer over the toy_text engines, made by RUNNING THE REFERENCE's
gym.wrappers.RecordEpisodeStatistics (gym/wrappers/record_episode_statistics.py:96-151) in the build container:

    python tests/golden/make_golden_toytext_stats.py          -> tests/golden/toytext_stats_<tag>.npz

Per id (FrozenLake-v1, Taxi-v3, Blackjack-v1) the same seeded trajectory is run twice:
  (1) RecordEpisodeStatistics(gym.vector.make(id, 8))                 the vector-level wrapper: infos["episode"] = {"r", "l", "t"} float64
                                                                      arrays of length N + the "_episode" mask (:10-37, :124-143);
  (2) gym.vector.make(id, 8, wrappers=RecordEpisodeStatistics)        the wrapper around every sub-env (gym/vector/__init__.py:56-65):
                                                                      infos["final_info"][i]["episode"] = {"r": float32, "l": int32, "t"}
with every random draw of every sub-env recorded (np_random.random() of the tabular envs, np_random.choice(deck) of Blackjack), so the
engines can be fed the same uniforms / cards and must then report the same episode returns and lengths bit for bit."""
import os
import sys

import numpy as np

for _name, _val in (("bool8", np.bool_), ("float_", np.float64), ("alltrue", np.all)):
    if not hasattr(np, _name):
        setattr(np, _name, _val)
sys.path.insert(0, "/root/reference")
import warnings  # noqa: E402

import gym  # noqa: E402
from gym.wrappers import RecordEpisodeStatistics  # noqa: E402

warnings.filterwarnings("ignore")
gym.logger.set_level(gym.logger.ERROR)
HERE = os.path.dirname(os.path.abspath(__file__))
N, MAX_DRAWS = 8, 24
CASES = {"FrozenLake-v1": ("FrozenLake-v1", {}, 160), "Taxi-v3": ("Taxi-v3", {}, 430),
         "Blackjack-v1": ("Blackjack-v1", {"natural": True, "sab": False}, 200)}


class Recorder:
    def __init__(self, g):
        self.g, self.uniforms, self.cards = g, [], []

    def random(self, *a, **k):
        v = self.g.random(*a, **k)
        self.uniforms.append(float(v))
        return v

    def choice(self, a, *args, **kw):
        v = self.g.choice(a, *args, **kw)
        if isinstance(a, list) and len(a) == 13:
            self.cards.append(int(v))
        return v

    def __getattr__(self, name):
        return getattr(self.g, name)


def run(gid, kw, T, per_sub_env):
    blackjack = gid == "Blackjack-v1"
    if per_sub_env:
        venv = gym.vector.make(gid, num_envs=N, asynchronous=False, wrappers=RecordEpisodeStatistics, **kw)
        top = venv
    else:
        venv = gym.vector.make(gid, num_envs=N, asynchronous=False, **kw)
        top = RecordEpisodeStatistics(venv)
    raws = [e.unwrapped for e in venv.envs]
    venv.action_space.seed(31)
    top.reset(seed=777)
    for r in raws:
        r._np_random = Recorder(r._np_random)
    obs0, _ = top.reset()                      # a second, recorded reset: the engines start from these draws
    first = [list(r._np_random.cards[:4]) if blackjack else list(r._np_random.uniforms[:1]) for r in raws]
    rec = {k: [] for k in ("actions", "draws", "ndraws", "reward", "terminated", "truncated", "ep_r", "ep_l", "ep_mask")}
    for t in range(T):
        a = venv.action_space.sample()
        for r in raws:
            r._np_random.uniforms.clear(), r._np_random.cards.clear()
        o, rw, te, tr, info = top.step(a)
        te, tr, rw = np.asarray(te, dtype=bool), np.asarray(tr, dtype=bool), np.asarray(rw)   # (the vector-level wrapper hands back lists, :121-122)
        done = te | tr
        d = np.zeros((N, MAX_DRAWS), np.float64)
        nd = np.zeros(N, np.int32)
        for i, r in enumerate(raws):
            log = r._np_random.cards if blackjack else r._np_random.uniforms
            d[i, :len(log)], nd[i] = log, len(log)
        er, el = np.zeros(N, np.float64), np.zeros(N, np.float64)
        if per_sub_env:
            assert "episode" not in info
            if "final_info" in info:
                for i, fi in enumerate(info["final_info"]):
                    if fi is not None:
                        ep = fi["episode"]
                        assert isinstance(ep["r"], np.float32) and isinstance(ep["l"], np.int32), ep
                        er[i], el[i] = ep["r"], ep["l"]
            mask = done.copy()
        else:
            mask = np.zeros(N, bool)
            if "episode" in info:
                ep = info["episode"]
                assert ep["r"].dtype == np.float64 and ep["l"].dtype == np.float64 and set(ep) == {"r", "l", "t"}, ep
                er, el, mask = ep["r"].copy(), ep["l"].copy(), info["_episode"].copy()
            assert np.array_equal(mask, done)
        for k, v in (("actions", a), ("draws", d), ("ndraws", nd), ("reward", rw), ("terminated", te), ("truncated", tr), ("ep_r", er), ("ep_l", el),
                     ("ep_mask", mask)):
            rec[k].append(v)
    out = {k: np.stack(v) for k, v in rec.items()}
    out["first"] = np.array(first, np.float64)
    return out


def main():
    for tag, (gid, kw, T) in CASES.items():
        a, b = run(gid, kw, T, False), run(gid, kw, T, True)
        for k in ("actions", "draws", "ndraws", "reward", "terminated", "truncated", "first", "ep_mask"):
            assert np.array_equal(a[k], b[k]), (tag, k)                 # the same trajectory, wrapped two ways
        assert np.array_equal(a["ep_l"], b["ep_l"]) and np.array_equal(a["ep_r"], b["ep_r"])   # float32 sums widened == float64 arrays
        limit = gym.spec(gid).max_episode_steps
        out = dict(a, sub_ep_r=b["ep_r"].astype(np.float32), sub_ep_l=b["ep_l"].astype(np.int32), max_episode_steps=np.int64(limit or -1),
                   natural=np.bool_(kw.get("natural", False)), sab=np.bool_(kw.get("sab", True)))
        path = os.path.join(HERE, f"toytext_stats_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"{tag}: T={T} episodes={int(a['ep_mask'].sum())} truncated={int(a['truncated'].sum())} returns={sorted(set(a['ep_r'][a['ep_mask']]))[:8]} -> {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
