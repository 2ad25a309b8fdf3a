#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8(f)-4 — the tabular toy_text envs — made by RUNNING THE REFERENCE (build container only):

    python tests/golden/make_golden_toytext.py          -> tests/golden/toytext_<Id>.npz

Per id: (1) the reference's own MDP table `env.P` and `initial_state_distrib`, dumped densely (what gym_amd.toy_text's
builders must reproduce entry for entry); (2) a trajectory of gym.vector.make(id, num_envs=8, asynchronous=False) with seeded
random actions in which every `np_random.random()` call of every sub-env is recorded, so the engine (and the oracle) can be
fed the very same uniforms and must then reproduce observations, rewards, flags, infos["prob"] (including the dtype the
reference's VectorEnv._add_info gives it), final_observation / final_info bit for bit.  Deterministic (fixed seeds).
"""
import os
import sys

import numpy as np

for _name, _val in (("bool8", np.bool_), ("float_", np.float64), ("alltrue", np.all)):
    if not hasattr(np, _name):
        setattr(np, _name, _val)

sys.path.insert(0, "/root/reference")
import warnings  # noqa: E402

import gym  # noqa: E402

warnings.filterwarnings("ignore")
gym.logger.set_level(gym.logger.ERROR)
HERE = os.path.dirname(os.path.abspath(__file__))

# tag: (gym id, make kwargs, steps)
CASES = {
    "FrozenLake-v1": ("FrozenLake-v1", {}, 150),
    "FrozenLake8x8-v1": ("FrozenLake8x8-v1", {}, 260),          # crosses the 200-step TimeLimit
    "FrozenLake-v1_deterministic": ("FrozenLake-v1", {"is_slippery": False}, 60),
    "FrozenLake-v1_limit7": ("FrozenLake-v1", {"max_episode_steps": 7}, 60),   # truncation-heavy
    "Taxi-v3": ("Taxi-v3", {}, 420),                              # crosses the 200-step TimeLimit twice
    "CliffWalking-v0": ("CliffWalking-v0", {}, 400),              # no TimeLimit; termination only at the goal
}
N = 8


class Recorder:
    """np_random stand-in that logs every .random() (the only generator call on this path, utils.py:8)."""

    def __init__(self, g):
        self.g, self.log = g, []

    def random(self, *a, **k):
        v = self.g.random(*a, **k)
        self.log.append(float(v))
        return v

    def __getattr__(self, name):
        return getattr(self.g, name)


def dump_table(env):
    S, A = env.observation_space.n, env.action_space.n
    M = max(len(env.P[s][a]) for s in range(S) for a in range(A))
    cnt = np.zeros((S, A), np.int32)
    prob = np.zeros((S, A, M))
    nxt = np.zeros((S, A, M), np.int32)
    rew = np.zeros((S, A, M))
    term = np.zeros((S, A, M), np.uint8)
    for s in range(S):
        for a in range(A):
            tr = env.P[s][a]
            cnt[s, a] = len(tr)
            for i, (p, ns, r, t) in enumerate(tr):
                prob[s, a, i], nxt[s, a, i], rew[s, a, i], term[s, a, i] = p, ns, r, t
    out = dict(num_transitions=cnt, prob=prob, next_state=nxt, reward=rew, terminated=term,
               initial_distrib=np.asarray(env.initial_state_distrib, np.float64))
    if hasattr(env, "action_mask"):
        out["action_mask"] = np.stack([env.action_mask(s) for s in range(S)]).astype(np.int8)
    return out


def main():
    for tag, (gid, kw, T) in CASES.items():
        venv = gym.vector.make(gid, num_envs=N, asynchronous=False, **kw)
        raws = [e.unwrapped for e in venv.envs]
        table = dump_table(raws[0])
        limit = venv.envs[0].spec.max_episode_steps if "max_episode_steps" not in kw else kw["max_episode_steps"]
        venv.action_space.seed(99)
        obs0, info0 = venv.reset(seed=2024)
        for r in raws:
            r._np_random = Recorder(r._np_random)
        A = raws[0].action_space.n
        acts = np.zeros((T, N), np.int64)
        u = np.full((T, 2, N), 0.5)
        obs = np.zeros((T, N), np.int64)
        rew = np.zeros((T, N))
        term = np.zeros((T, N), np.bool_)
        trunc = np.zeros((T, N), np.bool_)
        prob = np.zeros((T, N))
        prob_is_int = np.zeros(T, np.bool_)
        fin_obs = np.zeros((T, N), np.int64)
        fin_mask = np.zeros((T, N), np.bool_)
        fin_prob = np.zeros((T, N))
        has_mask = "action_mask" in table
        amask = np.zeros((T, N, A), np.int8)
        for t in range(T):
            a = venv.action_space.sample()
            if gid == "CliffWalking-v0" and t % 50 < 15:   # random walks never reach the goal: UP, 11x RIGHT, 3x DOWN does
                a = np.full(N, ([0] + [1] * 11 + [2] * 3)[t % 50], dtype=a.dtype)
            for r in raws:
                r._np_random.log.clear()
            o, rw, te, tr, info = venv.step(a)
            acts[t], obs[t], rew[t], term[t], trunc[t] = a, o, rw, te, tr
            assert o.dtype == np.int64 and rw.dtype == np.float64
            for i, r in enumerate(raws):
                log = r._np_random.log
                assert len(log) == 1 + int(te[i] or tr[i]), (log, te[i], tr[i])
                u[t, 0, i] = log[0]
                if len(log) > 1:
                    u[t, 1, i] = log[1]
            assert info["_prob"].all()
            prob[t] = info["prob"]
            prob_is_int[t] = np.issubdtype(info["prob"].dtype, np.integer)
            if has_mask:
                assert info["action_mask"].dtype == object
                amask[t] = np.stack(list(info["action_mask"]))
            if "final_observation" in info:
                assert info["final_observation"].dtype == np.int64       # python ints -> an int array, not objects
                fin_mask[t] = info["_final_observation"]
                fin_obs[t] = info["final_observation"]
                for i in np.flatnonzero(fin_mask[t]):
                    fin_prob[t, i] = info["final_info"][i]["prob"]
                    assert set(info["final_info"][i]) == ({"prob", "action_mask"} if has_mask else {"prob"})
            assert np.array_equal(fin_mask[t], te | tr)
        out = os.path.join(HERE, f"toytext_{tag}.npz")
        np.savez_compressed(out, id=gid, is_slippery=np.bool_(kw.get("is_slippery", True)),
                            max_episode_steps=np.int64(-1 if limit is None else limit), obs0=obs0,
                            reset_prob_dtype_is_int=np.bool_(np.issubdtype(info0["prob"].dtype, np.integer)),
                            actions=acts, uniforms=u, obs=obs, reward=rew, terminated=term, truncated=trunc, prob=prob,
                            prob_is_int=prob_is_int, final_obs=fin_obs, final_mask=fin_mask, final_prob=fin_prob,
                            step_action_mask=amask, **{f"table_{k}": v for k, v in table.items()})
        print(f"{tag}: T={T} done={int(fin_mask.sum())} trunc={int(trunc.sum())} int-prob steps={int(prob_is_int.sum())} "
              f"-> {os.path.getsize(out)} B")


if __name__ == "__main__":
    main()
