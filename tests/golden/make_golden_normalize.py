#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8(f)-2 — gym.wrappers.NormalizeObservation / NormalizeReward — made by RUNNING THE
REFERENCE's own wrappers (gym/wrappers/normalize.py:8-144) over its SyncVectorEnv, in the build container only:

    python tests/golden/make_golden_normalize.py          -> tests/golden/normalize_<Env>.npz

Two identically seeded SyncVectorEnvs are stepped with the same action tape: a bare one records the raw observations /
rewards / flags (the inputs of the normalisation), one wrapped as NormalizeReward(NormalizeObservation(env)) records the
reference's normalised outputs and its final running statistics.  Deterministic (fixed seeds).
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
from gym.wrappers.normalize import NormalizeObservation, NormalizeReward  # noqa: E402

warnings.filterwarnings("ignore")
gym.logger.set_level(gym.logger.ERROR)
HERE = os.path.dirname(os.path.abspath(__file__))

# name: (gym id, num_envs, steps, gamma)
CASES = {
    "CartPole": ("CartPole-v1", 8, 160, 0.99),
    "Pendulum": ("Pendulum-v1", 16, 230, 0.99),     # crosses the 200-step TimeLimit: returns zeroed on truncation
    "Acrobot": ("Acrobot-v1", 5, 120, 0.9),
    "MountainCarContinuous": ("MountainCarContinuous-v0", 64, 40, 0.99),
}


def main():
    for name, (gid, n, T, gamma) in CASES.items():
        raw = gym.vector.make(gid, num_envs=n, asynchronous=False)
        wrapped = NormalizeReward(NormalizeObservation(gym.vector.make(gid, num_envs=n, asynchronous=False)), gamma=gamma)
        raw.action_space.seed(1234)
        o_raw, _ = raw.reset(seed=77)
        o_nrm, _ = wrapped.reset(seed=77)
        O = o_raw.shape[1]
        raw_obs = np.zeros((T + 1, n, O), np.float32)
        nrm_obs = np.zeros((T + 1, n, O), np.float64)
        raw_rew = np.zeros((T, n), np.float64)
        nrm_rew = np.zeros((T, n), np.float64)
        term = np.zeros((T, n), np.bool_)
        trunc = np.zeros((T, n), np.bool_)
        raw_obs[0], nrm_obs[0] = o_raw, o_nrm
        assert o_nrm.dtype == np.float64
        for t in range(T):
            a = raw.action_space.sample()
            o1, r1, te1, tr1, _ = raw.step(a)
            o2, r2, te2, tr2, _ = wrapped.step(a)
            assert np.array_equal(te1, te2) and np.array_equal(tr1, tr2)
            raw_obs[t + 1], nrm_obs[t + 1] = o1, o2
            raw_rew[t], nrm_rew[t], term[t], trunc[t] = r1, r2, te1, tr1
        obs_rms = wrapped.env.obs_rms
        ret_rms = wrapped.return_rms
        out = os.path.join(HERE, f"normalize_{name}.npz")
        np.savez_compressed(
            out, raw_obs=raw_obs, nrm_obs=nrm_obs, raw_rew=raw_rew, nrm_rew=nrm_rew, term=term, trunc=trunc,
            gamma=np.float64(gamma), obs_epsilon=np.float64(wrapped.env.epsilon), rew_epsilon=np.float64(wrapped.epsilon),
            obs_mean=obs_rms.mean, obs_var=obs_rms.var, obs_count=np.float64(obs_rms.count),
            ret_mean=np.float64(ret_rms.mean), ret_var=np.float64(ret_rms.var), ret_count=np.float64(ret_rms.count),
            returns=wrapped.returns)
        print(f"{name}: N={n} T={T} done={int((term | trunc).sum())} -> {out} ({os.path.getsize(out)} B)")


if __name__ == "__main__":
    main()
