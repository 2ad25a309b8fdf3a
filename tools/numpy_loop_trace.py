#!/usr/bin/env python3
"""HipVectorEnv.step at 2^20 CartPole envs, 40 steps with sampled actions: run under
`rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats` to see where a step's 1.2 ms go (API calls, DMA copies, kernels)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gym_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
env = gym_amd.make("CartPole-v1", num_envs=n)
env.reset(seed=0)
env.action_space.seed(0)
acts = [env.action_space.sample() for _ in range(4)]
for i in range(30):
    env.step(acts[i % 4])
h = env._handle
t0 = time.perf_counter()
for i in range(steps):
    out = env.step(acts[i % 4])
dt = (time.perf_counter() - t0) / steps * 1e6
# the C-ABI call alone, then with / without the final-observation rows
t0 = time.perf_counter()
for i in range(steps):
    r = h.step_host(acts[i % 4], want_final=True, pooled=True)
    c = int(h._packed_views[0][0])
dt_native = (time.perf_counter() - t0) / steps * 1e6
h.final_packed(False)
t0 = time.perf_counter()
for i in range(steps):
    r = h.step_host(acts[i % 4], want_final=False, pooled=True)
dt_nofinal = (time.perf_counter() - t0) / steps * 1e6
print(json.dumps({"num_envs": n, "env.step_us": round(dt, 1), "step_host_packed_us": round(dt_native, 1), "finished_last": c,
                  "step_host_nofinal_us": round(dt_nofinal, 1)}))
env.close()
