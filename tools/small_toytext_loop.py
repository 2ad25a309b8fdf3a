import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import gym_amd
for gid in ("FrozenLake-v1", "Taxi-v3", "Blackjack-v1", "CartPole-v1"):
    for n in (8, 64):
        env = gym_amd.make(gid, num_envs=n)
        env.reset(seed=0)
        env.action_space.seed(0)
        acts = [env.action_space.sample() for _ in range(3)]
        for i in range(20):
            env.step(acts[i % 3])
        t0 = time.perf_counter()
        for i in range(2000):
            env.step(acts[i % 3])
        print(json.dumps({"id": gid, "num_envs": n, "us_per_step": round((time.perf_counter() - t0) / 2000 * 1e6, 1)}), flush=True)
        env.close()
