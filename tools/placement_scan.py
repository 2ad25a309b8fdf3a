#!/usr/bin/env python3
"""Placement scan inside ONE device allocation (continued).  The fused CartPole rollout runs in a fast or a slow
mode depending on where its five trajectory tensors sit relative to each other.  Separate allocations cannot tell
whether that is a property of the ADDRESSES (then a layout rule exists) or of the physical pages behind them.  Here all
candidates are carved out of the same 14-GiB block — same physical memory for every candidate — at systematically varied relative
offsets; the block is touched once, then every layout is timed twice, in shuffled order.

Box check first: boxes whose write path tops out at ~5 TB/s have no fast mode at all (profiles/r02a_placement_single_block.jsonl);
on those the scan only records the box kind and exits.

    python tools/placement_scan.py [--force]"""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--chunk", type=int, default=256)
    args = ap.parse_args()
    import torch
    from gym_amd.rollout import DeviceRollout

    n, K = args.n, args.chunk
    r = DeviceRollout("CartPole-v1", n, seed=0, action_seed=1)
    r.reset(seed=0)
    dev = r.device
    sizes = {"obs": K * n * 16, "reward": K * n * 8, "actions": K * n * 8, "terminated": K * n, "truncated": K * n}
    order = ["obs", "reward", "actions", "terminated", "truncated"]
    GiB = 1 << 30
    with torch.cuda.stream(r.stream):
        block = torch.empty(14 * GiB, dtype=torch.uint8, device=dev)
        block.zero_()
    r.synchronize()
    base0 = (-block.data_ptr()) % (2 << 20)

    def views(offs):
        o = {}
        for name in order:
            a = base0 + offs[name]
            b = block[a:a + sizes[name]]
            if name == "obs":
                o[name] = b.view(torch.float32).view(K, n, 4)
            elif name == "reward":
                o[name] = b.view(torch.float64).view(K, n)
            elif name == "actions":
                o[name] = b.view(torch.int64).view(K, n)
            else:
                o[name] = b.view(K, n)
        return o

    def timed(traj, launches=6):
        for _ in range(2):
            r.rollout_per_step(K, out=traj)
        r.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(r.stream)
        for _ in range(launches):
            r.rollout_per_step(K, out=traj)
        e1.record(r.stream)
        r.synchronize()
        return round(e0.elapsed_time(e1) / launches / K * 1e3, 3)

    def packed(gap=0, align=4096, perm=order, start=0):
        offs, off = {}, start
        for name in perm:
            off = (off + align - 1) // align * align
            offs[name] = off
            off += sizes[name] + gap
        return offs

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        r.rollout_per_step(K, out=views(packed()))
        r.synchronize()
    probe = [timed(views(packed(start=s * GiB))) for s in (0, 1, 2, 3, 0, 1)]
    kind = "fast" if min(probe) < 6.2 else "slow"
    print(json.dumps({"box": kind, "probe_us_per_step": probe}), flush=True)
    if kind == "slow" and not args.force:
        return
    layouts = {}
    for g in (0, 4 << 10, 64 << 10, 68 << 10, 1 << 20, (2 << 20) + (68 << 10), 33 << 20, 257 << 20):
        layouts[f"packed gap={g >> 10}KiB"] = packed(gap=g)
    for al in (2 << 20, 64 << 20, 1 << 30):
        layouts[f"aligned {al >> 20}MiB"] = packed(align=al)
    for s in (0, 1, 2, 3, 4):
        layouts[f"packed start={s}GiB"] = packed(start=s * GiB + s * (4 << 10))
    layouts["order flags first"] = packed(perm=["terminated", "truncated", "obs", "reward", "actions"])
    layouts["order reward obs actions"] = packed(perm=["reward", "obs", "actions", "terminated", "truncated"])
    layouts["order actions last, flags between"] = packed(perm=["obs", "terminated", "reward", "truncated", "actions"])
    rng = random.Random(1)
    for i in range(8):   # random order, random 4-KiB-aligned gaps (the five tensors are 8.5 GiB; up to 5 GiB of gaps in the 14-GiB block)
        perm = order[:]
        rng.shuffle(perm)
        cuts = sorted(rng.randrange(0, 5 * GiB // 4096) for _ in range(5))
        gaps = [cuts[0]] + [cuts[j] - cuts[j - 1] for j in range(1, 5)]
        offs, off = {}, 0
        for name, g in zip(perm, gaps):
            off += g * 4096
            offs[name] = off
            off += sizes[name]
        layouts[f"random {i}"] = offs
    names = list(layouts)
    res = {k: [] for k in names}
    for rep in range(2):
        rng.shuffle(names)
        for k in names:
            res[k].append(timed(views(layouts[k])))
    for k in layouts:
        print(json.dumps({"layout": k, "us_per_step": res[k], "offsets_MiB": {a: round(b / (1 << 20), 3) for a, b in layouts[k].items()}}), flush=True)
    r.close()


if __name__ == "__main__":
    main()
