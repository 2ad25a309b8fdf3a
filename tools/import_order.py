#!/usr/bin/env python3
"""gym_amd first, torch afterwards, in one process: both must see the GPU (one HIP runtime: gym_amd/_native.py,
_share_torch_hip_runtime)."""
import os
import sys
import time

T0 = time.time()


def stamp(what):
    print(f"[import_order] {what}: {time.time() - T0:.1f} s", file=sys.stderr, flush=True)


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gym_amd

assert "torch" not in sys.modules
stamp("import gym_amd")
env = gym_amd.make("CartPole-v1", num_envs=4096)
stamp("make")
env.reset(seed=0)
env.action_space.seed(0)
for _ in range(5):
    obs, rew, te, tr, infos = env.step(env.action_space.sample())
assert "torch" not in sys.modules
stamp("steps")
import torch

stamp("import torch")

assert torch.cuda.is_available(), "torch lost the GPU"
x = torch.ones(1 << 20, device="cuda")
assert float(x.sum()) == float(1 << 20)
stamp("torch sum")
norm = gym_amd.NormalizeObservation(gym_amd.make("CartPole-v1", num_envs=4096))
norm.reset(seed=0)
o = norm.step(norm.action_space.sample())[0]
assert o.dtype == np.float64 and np.isfinite(o).all()
stamp("normalize wrapper")
obs, rew, te, tr, infos = env.step(env.action_space.sample())
print("ok: gym_amd first, torch second; hip runtime:", torch.version.hip)
