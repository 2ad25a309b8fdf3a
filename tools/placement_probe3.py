#!/usr/bin/env python3
"""Placement autotune feasibility: hold several candidate sets of trajectory tensors at once (so each sits on different
physical pages), time the fused rollout on each, then re-time the best and the worst: is the speed a stable property of a set?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_amd.rollout import DeviceRollout

n, K = 1 << 20, 256
r = DeviceRollout("CartPole-v1", n, seed=0, action_seed=1)
r.reset(seed=0)


def timed(traj, launches):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(r.stream)
    for _ in range(launches):
        r.rollout_per_step(K, out=traj)
    e1.record(r.stream)
    r.synchronize()
    return e0.elapsed_time(e1) / launches / K * 1e3


sets = []
for i in range(12):
    traj = r.trajectory_buffers(K)
    timed(traj, 3)
    sets.append((timed(traj, 6), traj))
print("first pass:", " ".join(f"{t:.2f}" for t, _ in sets), flush=True)
sets = [(timed(tr, 6), tr) for _, tr in sets]
print("second pass:", " ".join(f"{t:.2f}" for t, _ in sets), flush=True)
# which tensor carries the effect?  start from the best set, swap in ONE tensor of the worst set at a time
order = sorted(range(len(sets)), key=lambda i: sets[i][0])
best, worst = sets[order[0]][1], sets[order[-1]][1]
for key in ("obs", "reward", "actions", "terminated", "truncated"):
    mix = dict(best)
    mix[key] = worst[key]
    print("best set with the worst set's", key, f"{timed(mix, 6):.2f}", flush=True)
order = sorted(range(len(sets)), key=lambda i: sets[i][0])
for label, i in (("best", order[0]), ("worst", order[-1]), ("median", order[len(order) // 2])):
    print(label, i, " ".join(f"{timed(sets[i][1], 8):.2f}" for _ in range(3)), flush=True)
r.close()
