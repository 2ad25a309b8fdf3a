#!/usr/bin/env python3
"""The NumPy loop with the reference-shaped wrappers stacked on HipVectorEnv: RecordEpisodeStatistics, NormalizeObservation,
NormalizeReward (us per step, PCIe + Python inclusive)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_amd


def loop(env, steps):
    env.reset(seed=0)
    env.action_space.seed(0)
    acts = [env.action_space.sample() for _ in range(4)]
    for i in range(6):
        env.step(acts[i % 4])
    t0 = time.perf_counter()
    for i in range(steps):
        out = env.step(acts[i % 4])
        infos = out[4]
        if "episode" in infos:
            infos["_episode"].any()
    dt = (time.perf_counter() - t0) / steps * 1e6
    env.close()
    return round(dt, 1)


for n, steps in ((4096, 500), (65536, 200), (1 << 20, 40)):
    row = {"num_envs": n}
    row["plain"] = loop(gym_amd.make("CartPole-v1", num_envs=n), steps)
    row["RecordEpisodeStatistics"] = loop(gym_amd.RecordEpisodeStatistics(gym_amd.make("CartPole-v1", num_envs=n)), steps)
    row["NormalizeObservation"] = loop(gym_amd.NormalizeObservation(gym_amd.make("CartPole-v1", num_envs=n)), steps)
    row["NormalizeReward"] = loop(gym_amd.NormalizeReward(gym_amd.make("CartPole-v1", num_envs=n)), steps)
    print(json.dumps(row), flush=True)
