#!/usr/bin/env python3
"""The gym-compatible NumPy loop (HipVectorEnv.step: NumPy actions in, NumPy obs / reward / flags / infos out; PCIe + Python
inclusive) at several sizes and copy modes.  Never the bench `value`; this is what a user who swaps gym.vector.SyncVectorEnv
for the engine and changes nothing else gets."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gym_amd


def loop(env_id, n, steps, keep=False, **kw):
    env = gym_amd.make(env_id, num_envs=n, **kw)
    env.reset(seed=0)
    env.action_space.seed(0)
    acts = [env.action_space.sample() for _ in range(4)]
    held = []
    for i in range(8):
        env.step(acts[i % 4])
    t0 = time.perf_counter()
    for i in range(steps):
        out = env.step(acts[i % 4])
        if keep:
            held.append(out)               # the caller keeps every result (a replay buffer of references): the pool cannot recycle
            if len(held) > 4:
                held.pop(0)
    dt = time.perf_counter() - t0
    env.close()
    return dt / steps * 1e6


for env_id in ("CartPole-v1", "Pendulum-v1"):
    for n, steps in ((8, 2000), (4096, 1000), (65536, 300), (1 << 20, 200)):
        for label, kw, keep in (("copy=True", {}, False), ("copy=True, caller keeps 4 results", {}, True), ("copy=False", dict(copy=False), False),
                                ("zero_copy=True", dict(zero_copy=True), False)):
            us = loop(env_id, n, steps, keep=keep, **kw)
            print(json.dumps({"env": env_id, "num_envs": n, "mode": label, "us_per_step": round(us, 1),
                              "env_steps_per_s": float(f"{n / us * 1e6:.4g}")}), flush=True)
