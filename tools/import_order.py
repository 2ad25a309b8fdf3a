#!/usr/bin/env python3
"""gym_amd first, torch afterwards, in one process: both must see the GPU (one HIP runtime: gym_amd/_native.py,
_share_torch_hip_runtime)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gym_amd

assert "torch" not in sys.modules
env = gym_amd.make("CartPole-v1", num_envs=4096)
env.reset(seed=0)
env.action_space.seed(0)
for _ in range(5):
    obs, rew, te, tr, infos = env.step(env.action_space.sample())
assert "torch" not in sys.modules
import torch

assert torch.cuda.is_available(), "torch lost the GPU"
x = torch.ones(1 << 20, device="cuda")
assert float(x.sum()) == float(1 << 20)
norm = gym_amd.NormalizeObservation(gym_amd.make("CartPole-v1", num_envs=4096))
norm.reset(seed=0)
o = norm.step(norm.action_space.sample())[0]
assert o.dtype == np.float64 and np.isfinite(o).all()
obs, rew, te, tr, infos = env.step(env.action_space.sample())
print("ok: gym_amd first, torch second; hip runtime:", torch.version.hip)
