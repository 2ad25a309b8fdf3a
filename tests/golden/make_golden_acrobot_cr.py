#!/usr/bin/env python3
"""Acrobot_p1_threshold_cr.npz — the reference's Acrobot step on the 4096 threshold states of Acrobot_p1_threshold.npz with a
CORRECTLY ROUNDED libm: the reference module's `sin` / `cos` (gym/envs/classic_control/acrobot.py:6 `from numpy import cos, pi, sin`)
are replaced by mpmath evaluations at 400 bits rounded once to float64; everything else is the reference's own code and arithmetic.

Why.  Within an ulp of the termination threshold (acrobot.py:235) the reference's mask depends on the last bit of its libm: glibc 2.35
(what NumPy's scalar cos / sin call here) returns the correctly rounded value for 99.88 % of arguments (measured: 369 of 310 000
differ from 400-bit mpmath) — and the engine's exact path (gym_amd/csrc/mxv_exact.hpp) returns it for all of them.  The engine's
contract in the band is therefore: the reference's arithmetic on a correctly rounded sin / cos.  This file is that, run by the reference
itself; tests/test_gpu_parity.py::test_acrobot_termination_threshold_states asserts the device's masks, states and observations equal
it for EVERY state, and that the masks differ from the glibc run (Acrobot_p1_threshold.npz) only where this file does too.

Run in the build container only (needs /root/reference and mpmath):   python tests/golden/make_golden_acrobot_cr.py
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (NumPy-2 shim + reference import + helpers)
from make_golden_goal import ref_step  # noqa: E402

gym = mg.gym
mp.mp.prec = 400


def _cr(fn):
    def f(x):
        a = np.asarray(x, dtype=np.float64)
        if a.ndim == 0:
            return np.float64(float(fn(mp.mpf(float(a)))))
        return np.array([float(fn(mp.mpf(float(v)))) for v in a.ravel()]).reshape(a.shape)
    return f


def main():
    import gym.envs.classic_control.acrobot as acro

    g = np.load(os.path.join(HERE, "Acrobot_p1_threshold.npz"))
    n = len(g["action"])
    raw = gym.make("Acrobot-v1", disable_env_checker=True).unwrapped
    raw.reset(seed=0)
    acro.cos, acro.sin = _cr(mp.cos), _cr(mp.sin)   # the module-level names AcrobotEnv's methods resolve at call time
    obs = np.zeros((n, 6), np.float32)
    rew = np.zeros(n)
    term = np.zeros(n, np.uint8)
    s1 = np.zeros((n, 4))
    for i in range(n):
        o, r, te, post = ref_step(raw, g["state0"][i], g["action"][i])
        obs[i], rew[i], term[i], s1[i] = o, r, te, post
    differ = term != g["terminated"]
    np.savez_compressed(os.path.join(HERE, "Acrobot_p1_threshold_cr.npz"), obs=obs, reward=rew, terminated=term, state1=s1)
    print(f"Acrobot P1[threshold, correctly rounded libm]: {n} steps, {int(term.sum())} terminations; masks that differ from the glibc run: "
          f"{int(differ.sum())} (|margin| in ulps of 1.0: {np.abs(g['margin'][differ]) / 2.0 ** -52}); post-step states equal to the glibc "
          f"run bit for bit: {(s1 == g['state1']).all(axis=1).mean():.4f}")


if __name__ == "__main__":
    main()
