#!/usr/bin/env python3
"""CartPole_p2_beyond.npz — the reference's CartPole stepped ON after it terminated (no reset in between): cartpole.py:169-184 pays 1.0
in the step the pole falls (`steps_beyond_terminated = 0`) and 0.0 in every later step that is still terminated (plus a one-time
logger.warn).  Unreachable under SyncVectorEnv's autoreset; reachable through the engine's MXV_FLAG_NO_AUTORESET (single-env semantics).
64 raw envs from states close to the thresholds, 14 steps each with random actions, every step recorded.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_cartpole_beyond.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

gym = mg.gym


def main(n=64, T=14, seed=20260925):
    rng = np.random.default_rng(seed)
    raw = gym.make("CartPole-v1", disable_env_checker=True).unwrapped
    raw.reset(seed=0)
    s0 = np.stack([rng.uniform(-2.45, 2.45, n), rng.uniform(-1.5, 1.5, n), rng.uniform(-0.215, 0.215, n), rng.uniform(-1.5, 1.5, n)], axis=1)
    s0[::4, 0] = rng.choice([-1, 1], len(s0[::4])) * rng.uniform(2.3, 2.4, len(s0[::4]))       # about to leave the track
    act = rng.integers(0, 2, (T, n)).astype(np.int64)
    obs = np.zeros((T, n, 4), np.float32)
    rew = np.zeros((T, n))
    term = np.zeros((T, n), np.uint8)
    post = np.zeros((T, n, 4))
    for i in range(n):
        mg.set_state(raw, "CartPole", s0[i], False)           # also steps_beyond_terminated = None
        for t in range(T):
            o, r, te, tr, info = raw.step(int(act[t, i]))
            obs[t, i], rew[t, i], term[t, i], post[t, i] = o, r, te, mg.get_state(raw)
    np.savez_compressed(os.path.join(HERE, "CartPole_p2_beyond.npz"), state0=s0, action=act, obs=obs, reward=rew, terminated=term, state_post=post)
    print(f"CartPole P2[beyond]: {n} envs x {T} steps, {int(term.any(axis=0).sum())} envs terminate, {int((rew == 0).sum())} zero rewards, "
          f"{int(((term == 1) & (rew == 1)).sum())} falls paid 1.0, {int(((term == 0) & (np.cumsum(term, axis=0) > 0)).sum())} un-terminated steps after a fall")


if __name__ == "__main__":
    main()
