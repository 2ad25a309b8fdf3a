#!/usr/bin/env python3
"""configs[0] (8 envs): where a HipVectorEnv.step goes — the C call alone vs the adapter around it."""
import cProfile
import json
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gym_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
env = gym_amd.make("CartPole-v1", num_envs=n)
env.reset(seed=0)
env.action_space.seed(0)
acts = [env.action_space.sample() for _ in range(16)]
h = env._handle


def t(fn, reps=3000):
    for _ in range(50):
        fn()
    t0 = time.perf_counter()
    for i in range(reps):
        fn()
    return round((time.perf_counter() - t0) / reps * 1e6, 2)


out = {"num_envs": n}
out["env.step"] = t(lambda: env.step(acts[0]))
out["env.step(sample())"] = t(lambda: env.step(env.action_space.sample()))
out["handle.step_host_block"] = t(lambda: h.step_host_block(acts[0], want_final=True))
out["handle.step_host"] = t(lambda: h.step_host(acts[0], want_final=True, pooled=True))
out["action_space.sample"] = t(lambda: env.action_space.sample())
io = h.host_io()
out["handle.step_mapped"] = t(lambda: h.step_mapped())
print(json.dumps(out))
pr = cProfile.Profile()
pr.enable()
for i in range(3000):
    env.step(acts[i % 16])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
