#!/usr/bin/env python3
"""<env>_p1_nonfinite.npz — the reference's step on NON-FINITE inputs: NaN / +-Inf Box actions (a diverged policy) and NaN / +-Inf state
components (what such an action leaves behind, or an injected state).  The reference validates neither; what comes out is decided by
how each clamp is written — np.clip propagates a NaN (pendulum.py:127,132, mountain_car.py:134,136), Python's `if x > hi`, max(x, lo),
min(x, hi) keep a NaN first operand (continuous_mountain_car.py:146-157, acrobot.py:399-415) — and an engine whose clamps are hardware
min / max would silently turn the NaN into a bound.  Same arrays as make_golden.make_p1, same replay (helpers.run_p1 with
NaN == NaN); +-Inf angles are left out (sin(inf) is NaN with an FP exception, the wrap loops of Acrobot never end on them).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_nonfinite.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

gym = mg.gym
NAN, INF = float("nan"), float("inf")


def cases(name, rng, n):
    """n (state, action, fresh) triples: finite draws with one or two entries replaced."""
    gid, S, O, nd, _ = mg.ENVS[name]
    s = mg.p1_states(name, rng, n)
    a = mg.p1_actions(name, rng, n)
    fresh = np.zeros(n, np.uint8)
    if name == "MountainCarContinuous":
        fresh[: n // 2] = 1
        s[n // 2:] = s[n // 2:].astype(np.float32)
    angle_cols = {"CartPole": [2], "Pendulum": [0], "Acrobot": [0, 1], "MountainCar": [0], "MountainCarContinuous": [0]}[name]
    for i in range(n):
        kind = i % 6
        col = int(rng.integers(0, S))
        if kind == 0 and not nd:
            a[i] = NAN
        elif kind == 1 and not nd:
            a[i] = INF if i % 12 == 1 else -INF
        elif kind in (2, 3):
            s[i, col] = NAN
        elif kind == 4 and col not in angle_cols:
            s[i, col] = INF if rng.integers(0, 2) else -INF
        # kind 5 (and the skipped combinations): finite control rows
    return s, a, fresh


def make(name, n=384, seed=20260924):
    gid, S, O, nd, _ = mg.ENVS[name]
    rng = np.random.default_rng(seed + sum(map(ord, name)))
    raw = gym.make(gid, disable_env_checker=True).unwrapped
    raw.reset(seed=0)
    s0, act, fresh = cases(name, rng, n)
    obs = np.zeros((n, O), np.float32)
    rew = np.zeros(n)
    term = np.zeros(n, np.uint8)
    s1 = np.zeros((n, S))
    with np.errstate(all="ignore"):
        for i in range(n):
            mg.set_state(raw, name, s0[i], bool(fresh[i]))
            a = act[i] if nd else np.array([act[i]], dtype=np.float32)
            o, r, te, tr, info = raw.step(a)
            obs[i], rew[i], term[i], s1[i] = o, r, te, mg.get_state(raw)
    np.savez_compressed(os.path.join(HERE, f"{name}_p1_nonfinite.npz"), state0=s0, action=act, fresh=fresh, obs=obs, reward=rew,
                        terminated=term, state1=s1)
    print(f"{name:24s} P1[nonfinite]: {n} steps, NaN in {int(np.isnan(s1).any(axis=1).sum())} post-step states, "
          f"{int(np.isnan(rew).sum())} NaN rewards, {int(term.sum())} terminations")


if __name__ == "__main__":
    for name in mg.ENVS:
        make(name)
