#!/usr/bin/env python3
"""Golden vectors for AcrobotEnv.step with torque_noise_max > 0 (gym/envs/classic_control/acrobot.py:202-205), made by
RUNNING THE REFERENCE (build container only):   python tests/golden/make_golden_acrobot_noise.py

The env's np_random is replaced by a stub whose uniform(low, high) returns a recorded value drawn in [low, high], so the
noise term is an INPUT of the vector (the engine's own noise comes from its Philox step-noise stream; what is pinned here
is the arithmetic: torque = AVAIL_TORQUE[a] + noise, then the unchanged RK4 step)."""
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


class Stub:
    def __init__(self, value):
        self.value = value

    def uniform(self, low, high):
        assert low <= self.value <= high
        return self.value


def main():
    rng = np.random.default_rng(7)
    M, noise_max = 400, 0.4
    raw = gym.make("Acrobot-v1").unwrapped
    raw.reset(seed=0)
    raw.torque_noise_max = noise_max
    state = np.zeros((M, 4)); action = np.zeros(M, np.int64); noise = np.zeros(M)
    obs = np.zeros((M, 6), np.float32); reward = np.zeros(M); term = np.zeros(M, np.bool_); state_post = np.zeros((M, 4))
    for i in range(M):
        s = np.array([rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi), rng.uniform(-12, 12), rng.uniform(-28, 28)])
        if i % 5 == 0:
            s = rng.uniform(-0.1, 0.1, 4).astype(np.float32).astype(np.float64)   # reset-like states
        a = int(rng.integers(0, 3))
        v = float(rng.uniform(-noise_max, noise_max))
        raw.state = s.copy()
        raw._np_random = Stub(v)
        o, r, te, tr, _ = raw.step(a)
        state[i], action[i], noise[i] = s, a, v
        obs[i], reward[i], term[i], state_post[i] = o, r, te, np.asarray(raw.state, np.float64)
    out = os.path.join(HERE, "Acrobot_noise_p1.npz")
    np.savez_compressed(out, torque_noise_max=np.float64(noise_max), state=state, action=action, noise=noise, obs=obs,
                        reward=reward, terminated=term, state_post=state_post)
    print(f"{M} steps, {int(term.sum())} terminated -> {out} ({os.path.getsize(out)} B)")


if __name__ == "__main__":
    main()
