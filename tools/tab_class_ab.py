#!/usr/bin/env python3
"""tools/tab_class_ab.py — does the tabular engine's trajectory launch (four 8-B/lane streams obs / actions / reward / prob + two flag bytes:
34 B per env-step, three speed modes seen in round 1) care about HBM classes, and which streams must be kept apart?
Trajectory tensors are built with mxv_placed_alloc under different groupings; FrozenLake-v1, 2^20 envs, 128-step launches.  JSON lines."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_amd import _native  # noqa: E402
from gym_amd.toy_text import TabularRollout  # noqa: E402

N, K = 1 << 20, 128
r = TabularRollout("FrozenLake-v1", N, seed=0, action_seed=1)
r.reset(seed=0)
names = ["obs", "actions", "reward", "prob", "terminated", "truncated"]
dts = {"obs": "<i8", "actions": "<i8", "reward": "<f8", "prob": "<f8", "terminated": "|u1", "truncated": "|u1"}


def timed(out, reps=8):
    for _ in range(3):
        r.rollout_per_step(K, out=out)
    r.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(r.stream)
        for _ in range(reps):
            r.rollout_per_step(K, out=out)
        e1.record(r.stream)
        r.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps / K * 1e3)
    return round(best, 3)


warm = r.trajectory_buffers(K)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.5:
    r.rollout_per_step(K, out=warm)
    r.synchronize()
print(json.dumps({"grouping": "ordinary allocations (torch)", "us_per_step": timed(warm)}), flush=True)
del warm
torch.cuda.empty_cache()
groupings = {
    "obs+actions | reward+prob": {"obs": 0, "actions": 0, "reward": 1, "prob": 1},
    "obs | actions+reward+prob": {"obs": 0, "actions": 1, "reward": 1, "prob": 1},
    "obs+reward | actions+prob": {"obs": 0, "reward": 0, "actions": 1, "prob": 1},
    "obs+actions+reward | prob": {"obs": 0, "actions": 0, "reward": 0, "prob": 1},
}
for label, g in groupings.items():
    mem = _native.PlacedMemory(0, [(n, (K, N), dts[n], g.get(n, -1)) for n in names])
    out = mem.tensors()
    print(json.dumps({"grouping": label, "balanced": mem.info["balanced"], "class_chunks": mem.info["class_chunks"], "jumped_GiB": mem.info["jumped_GiB"],
                      "us_per_step": timed(out)}), flush=True)
    del out
    mem.close()
# everything on ONE class: group 0 only would take ordinary allocations, so: all in group 0 except a token 256-MiB tensor in group 1
mem = _native.PlacedMemory(0, [(n, (K, N), dts[n], 0 if n in ("obs", "actions", "reward", "prob") else -1) for n in names] + [("token", (256 << 20,), "|u1", 1),
                                                                                                                        ("pad", (2 << 30,), "|u1", -1)])
out = {k: v for k, v in mem.tensors().items() if k in names}
print(json.dumps({"grouping": "all four streams on one class", "balanced": mem.info["balanced"], "class_chunks": mem.info["class_chunks"], "us_per_step": timed(out)}), flush=True)
