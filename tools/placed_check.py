#!/usr/bin/env python3
"""tools/placed_check.py — does trajectory_buffers(layout="placed") (mxv_placed_alloc) give the fused CartPole rollout its fast mode in
every fresh process?  One JSON line per process: the placement report, the write probe and the fused rollout on the placed set, and the
same two figures on `--separate` ordinary (torch / hipMalloc) sets allocated one after the other and held, for the box's own spread.
--preamble N first allocates, touches and frees N GiB through torch (a process that has already used the device).
--check: the rollout into placed tensors equals the rollout into ordinary tensors bit for bit (same seeds)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=1 << 20)
ap.add_argument("--chunk", type=int, default=256)
ap.add_argument("--separate", type=int, default=4)
ap.add_argument("--preamble", type=float, default=0.0)
ap.add_argument("--check", action="store_true")
ap.add_argument("--env-id", default="CartPole-v1")
ap.add_argument("--layout", default="placed", choices=["placed", "sorted", "both"])
ap.add_argument("--warm-s", type=float, default=0.2, help="seconds of the workload before anything is measured (a cold box needs > 1 s)")
args = ap.parse_args()

import torch  # noqa: E402

from gym_amd import _native  # noqa: E402
from gym_amd.rollout import DeviceRollout  # noqa: E402

if args.preamble > 0:
    hold = [torch.zeros(int(args.preamble * 2**30 / 4) // 4, dtype=torch.float32, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    del hold
    torch.cuda.empty_cache()

r = DeviceRollout(args.env_id, args.envs, seed=0, action_seed=1)
r.reset(seed=0)
K = args.chunk


def timed(traj, launches=6):
    for _ in range(2):
        r.rollout_per_step(K, out=traj)
    r.stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(r.stream)
    for _ in range(launches):
        r.rollout_per_step(K, out=traj)
    e1.record(r.stream)
    r.stream.synchronize()
    return e0.elapsed_time(e1) / launches / K * 1e3


def probe(traj):
    if args.env_id != "CartPole-v1" or args.envs % 1024:
        return None
    torch.cuda.synchronize()
    return _native.write_probe(0, args.envs, K, 6, traj["obs"], traj["reward"], traj["actions"], traj["terminated"], traj["truncated"])


# clock ramp on a throw-away set
warm = r.trajectory_buffers(min(K, 64), layout="separate")
t0 = time.perf_counter()
while time.perf_counter() - t0 < args.warm_s:
    r.rollout_per_step(min(K, 64), out=warm)
    r.stream.synchronize()
del warm
torch.cuda.empty_cache()

out = {"layout": args.layout, "env_id": args.env_id, "envs": args.envs, "K": K, "preamble_GiB": args.preamble, "warm_s": args.warm_s}
for lay in (["placed", "sorted"] if args.layout == "both" else [args.layout]):
    t0 = time.perf_counter()
    placed = r.trajectory_buffers(K, layout=lay)
    t_alloc = time.perf_counter() - t0
    o = {"placement": r.last_placement, "alloc_s": round(t_alloc, 3), "rollout_us": round(timed(placed), 3)}
    p = probe(placed)
    if p is not None:
        o["probe_us"] = round(p, 3)
        o["rollout_over_probe"] = round(o["rollout_us"] / p, 3)
    o["rollout_us_again"] = round(timed(placed), 3)
    if args.layout == "both":
        out[lay] = o
        del placed
        torch.cuda.empty_cache()
    else:
        out.update({"placement": o["placement"], "alloc_s": o["alloc_s"], "placed_rollout_us": o["rollout_us"], "placed_probe_us": o.get("probe_us"),
                    "placed_rollout_over_probe": o.get("rollout_over_probe")})
sets, sep = [], []
for i in range(args.separate):
    s = r.trajectory_buffers(K, layout="separate")
    sets.append(s)
    p = probe(s)
    sep.append({"rollout_us": round(timed(s), 3), "probe_us": None if p is None else round(p, 3)})
out["separate_sets"] = sep
if args.check:
    a = DeviceRollout(args.env_id, 1 << 16, seed=3, action_seed=4)
    b = DeviceRollout(args.env_id, 1 << 16, seed=3, action_seed=4)
    a.reset(seed=3)
    b.reset(seed=3)
    ta = a.trajectory_buffers(K, layout="placed")
    tb = b.trajectory_buffers(K, layout="separate")
    for _ in range(2):
        a.rollout_per_step(K, out=ta)
        b.rollout_per_step(K, out=tb)
    a.synchronize()
    b.synchronize()
    out["check"] = {"placement": a.last_placement, "equal": all(bool(torch.equal(ta[k], tb[k])) for k in tb)}
print(json.dumps(out), flush=True)
