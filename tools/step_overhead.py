#!/usr/bin/env python3
"""Host-side cost of one DeviceRollout.step(actions) (learner-in-the-loop: one launch per step with caller actions): a small
vector env makes the kernel negligible, so the loop time is Python + ctypes + HIP launch overhead."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd.rollout import DeviceRollout

out = {}
for n in (1024, 1 << 20):
    r = DeviceRollout("CartPole-v1", n, seed=0, action_seed=1)
    r.reset(seed=0)
    acts = r.sample_actions().clone()
    torch.cuda.synchronize()
    steps = 5000 if n == 1024 else 2000

    def timed(fn):
        fn(); fn()
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        cpu = (time.perf_counter() - t0) / steps * 1e6
        r.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e6
        return {"enqueue_us": round(cpu, 2), "wall_us": round(wall, 2)}

    out[f"step_{n}"] = timed(lambda: r.step(acts, want_final=False))
    with torch.cuda.stream(r.stream):     # caller already on the engine's stream: no cross-stream wait needed
        out[f"step_on_engine_stream_{n}"] = timed(lambda: r.step(acts, want_final=False))
    out[f"raw_handle_step_{n}"] = timed(lambda: r.handle.step(acts, r.obs, r.reward, r.terminated, r.truncated, None))
    out[f"step_sampled_{n}"] = timed(lambda: r.step_sampled())
    r.close()
print(json.dumps(out, indent=1))
