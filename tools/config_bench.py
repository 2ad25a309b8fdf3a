#!/usr/bin/env python3
"""One-GPU numbers for every BASELINE.json config (bench.py only reports configs[1]): device-resident fused rollouts for
configs 2-5 at their per-GPU sizes, and the gym-compatible NumPy loop (HipVectorEnv.step, PCIe-inclusive) for
config 1 (num_envs=8, 1000 steps) and at 2^20 envs.  Prints one JSON line per measurement."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import gym_amd
from gym_amd.mixed import DEFAULT_MIX, MixedRollout
from gym_amd.rollout import DeviceRollout


def fused(env_id, n, steps=2000, chunk=100):
    r = DeviceRollout(env_id, n, seed=0, action_seed=1)
    r.reset(seed=0)
    traj = r.trajectory_buffers(chunk)   # sorted by HBM class
    for _ in range(30):
        r.rollout_per_step(chunk, out=traj)
    r.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(r.stream)
        for _ in range(steps // chunk):
            r.rollout_per_step(chunk, out=traj)
        e1.record(r.stream)
        r.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps * 1e3)
    r.close()
    return best


def compat_loop(env_id, n, steps, **kw):
    env = gym_amd.make(env_id, num_envs=n, **kw)
    env.reset(seed=0)
    env.action_space.seed(0)
    acts = [env.action_space.sample() for _ in range(8)]
    for i in range(5):
        env.step(acts[i % 8])
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 8])
    dt = time.perf_counter() - t0
    env.close()
    return n * steps / dt, dt / steps * 1e6


def main():
    out = []
    for cfg, env_id, n in (("config2", "CartPole-v1", 1 << 20), ("config3", "Pendulum-v1", 1 << 19),
                           ("config3", "MountainCarContinuous-v0", 1 << 19), ("config4 (per-GPU shard of 2^22)", "Acrobot-v1", 1 << 19)):
        us = fused(env_id, n)
        out.append({"config": cfg, "env": env_id, "num_envs": n, "mode": "device-resident fused rollout, per-step outputs",
                    "us_per_step": round(us, 3), "env_steps_per_s": float(f"{n / us * 1e6:.4g}")})
    # config 5: mixed batch, per-GPU share of 2^20 envs over 8 GPUs = 4 x 2^15
    mr = MixedRollout(1 << 17, DEFAULT_MIX, rank=0, world_size=1, device=0, seed=0, action_seed=1)
    mr.reset(seed=0)
    traj = mr.trajectory_buffers(100)
    for _ in range(20):
        mr.rollout_per_step(100, out=traj)
    mr.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        mr.rollout_per_step(100, out=traj)
    mr.synchronize()
    dt = time.perf_counter() - t0
    out.append({"config": "config5 (per-GPU share: 4 kinds x 2^15 envs, 4 streams)", "env": "+".join(DEFAULT_MIX), "num_envs": 1 << 17,
                "mode": "device-resident fused rollouts on 4 streams", "us_per_step": round(dt / 4000 * 1e6, 3),
                "env_steps_per_s": float(f"{(1 << 17) * 4000 / dt:.4g}")})
    mr.close()
    for cfg, n, steps in (("config1 (plumbing)", 8, 1000), ("gym-compatible loop at config2 size", 1 << 20, 30)):
        sps, us = compat_loop("CartPole-v1", n, steps)
        out.append({"config": cfg, "env": "CartPole-v1", "num_envs": n, "mode": "HipVectorEnv.step (NumPy in/out, PCIe + "
                    "Python inclusive)", "us_per_step": round(us, 1), "env_steps_per_s": float(f"{sps:.4g}")})
    # the same PCIe + Python inclusive loop for the other configs' env kinds at their per-GPU sizes (number iii)
    for cfg, env_id, n in (("config3", "Pendulum-v1", 1 << 19), ("config3", "MountainCarContinuous-v0", 1 << 19),
                           ("config4 (per-GPU shard of 2^22)", "Acrobot-v1", 1 << 19)):
        for label, kw in (("copies", {}), ("zero_copy=True", dict(zero_copy=True))):
            sps, us = compat_loop(env_id, n, 30, **kw)
            out.append({"config": f"gym-compatible loop, {cfg}, {label}", "env": env_id, "num_envs": n,
                        "mode": "HipVectorEnv.step (NumPy in/out, PCIe + Python inclusive)", "us_per_step": round(us, 1),
                        "env_steps_per_s": float(f"{sps:.4g}")})
    for label, kw in (("copy=False", dict(copy=False)), ("zero_copy=True", dict(zero_copy=True))):
        for n, steps in ((8, 1000), (1 << 20, 30)):
            sps, us = compat_loop("CartPole-v1", n, steps, **kw)
            out.append({"config": f"gym-compatible loop, {label}", "env": "CartPole-v1", "num_envs": n,
                        "mode": "HipVectorEnv.step, outputs are views of the pinned device-mapped I/O block",
                        "us_per_step": round(us, 1), "env_steps_per_s": float(f"{sps:.4g}")})
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
