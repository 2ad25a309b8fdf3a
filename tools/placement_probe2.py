#!/usr/bin/env python3
"""Follow-up to placement_probe.py: carve the five trajectory arrays out of ONE allocation with chosen byte skews between
them and see which relative placements are fast, and whether that is reproducible across fresh allocations."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd.rollout import DeviceRollout

n, K = 1 << 20, 256
r = DeviceRollout("CartPole-v1", n, seed=0, action_seed=1)
r.reset(seed=0)
SIZES = [("obs", K * n * 16, torch.float32, (K, n, 4)), ("reward", K * n * 8, torch.float64, (K, n)),
         ("actions", K * n * 8, torch.int64, (K, n)), ("terminated", K * n, torch.uint8, (K, n)), ("truncated", K * n, torch.uint8, (K, n))]
SKEWS = {
    "none": [0, 0, 0, 0, 0],
    "1k": [0, 1024, 2048, 3072, 4096],
    "c256": [0, 256, 256, 256, 256],
    "c512": [0, 512, 512, 512, 512],
    "c1k": [0, 1024, 1024, 1024, 1024],
    "c2k": [0, 2048, 2048, 2048, 2048],
    "c4k": [0, 4096, 4096, 4096, 4096],
    "c8k": [0, 8192, 8192, 8192, 8192],
}
junk = []
for trial in range(int(os.environ.get('TRIALS', '4'))):
    for name, sk in SKEWS.items():
        total = sum(s for _, s, _, _ in SIZES) + sum(sk) + (8 << 20)
        pool = torch.empty(total, dtype=torch.uint8, device="cuda")
        off, traj = 0, {}
        for (key, size, dt, shape), s in zip(SIZES, sk):
            off = (off + s + 255) // 256 * 256
            traj[key] = pool[off:off + size].view(dt).view(shape)
            off += size
        for _ in range(3):
            r.rollout_per_step(K, out=traj)
        r.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(r.stream)
        for _ in range(16):
            r.rollout_per_step(K, out=traj)
        e1.record(r.stream)
        r.synchronize()
        print(json.dumps({"trial": trial, "skew": name, "us_per_step": round(e0.elapsed_time(e1) / 16 / K * 1e3, 3),
                          "pool_MiB_mod_1G": (pool.data_ptr() % (1 << 30)) >> 20}), flush=True)
        del traj, pool
        torch.cuda.empty_cache()
        junk.append(torch.empty((17 + 13 * len(junk)) << 20, dtype=torch.uint8, device="cuda"))
r.close()
