#!/usr/bin/env python3
"""Is the 6.0-vs-7.1 us/step bimodality of the fused CartPole rollout a property of WHERE the trajectory tensors live?
One process, the same engine handle; the [256][N] trajectory tensors are re-allocated at shifted addresses between timings."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd.rollout import DeviceRollout

n, K = 1 << 20, 256
r = DeviceRollout("CartPole-v1", n, seed=0, action_seed=1)
r.reset(seed=0)
keep = []
for trial in range(10):
    if trial:
        keep.append(torch.empty((trial * 37 + 1) << 20, dtype=torch.uint8, device="cuda"))  # shift the next allocations
    traj = r.trajectory_buffers(K)
    for _ in range(4):
        r.rollout_per_step(K, out=traj)
    r.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(r.stream)
        for _ in range(20):
            r.rollout_per_step(K, out=traj)
        e1.record(r.stream)
        r.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 / K * 1e3)
    print(json.dumps({"trial": trial, "obs_ptr_mod_1GiB_MiB": (traj["obs"].data_ptr() % (1 << 30)) >> 20,
                      "rew_ptr_MiB": (traj["reward"].data_ptr() % (1 << 30)) >> 20, "us_per_step": [round(t, 3) for t in ts]}), flush=True)
    del traj
    torch.cuda.empty_cache()
r.close()
