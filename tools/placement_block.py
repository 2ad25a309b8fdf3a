#!/usr/bin/env python3
"""Placement experiment: do the five trajectory tensors of the fused CartPole rollout run in ONE speed mode
when they are carved out of ONE device allocation at fixed relative offsets, instead of five separate allocations whose
relative physical placement is a lottery (profiles/r01h_placement_probe.txt: 5.9 / 6.7 / 7.1 us per step)?

    python tools/placement_block.py --layout packed|sep|mib2|stagger|gib --compact 0|1

One process = one fresh set of allocations; the caller runs it several times per layout.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def carve(block, layout, K, n, O, rew_b, act_dt, rew_dt, torch):
    sizes = [("obs", K * n * O * 4), ("reward", K * n * rew_b), ("actions", K * n * act_dt.itemsize), ("terminated", K * n),
             ("truncated", K * n)]
    align = {"packed": 4096, "mib2": 2 << 20, "stagger": 2 << 20, "gib": 1 << 30}[layout]
    offs, off = {}, 0
    for j, (name, nb) in enumerate(sizes):
        off = (off + align - 1) // align * align
        if layout == "stagger":
            off += j * (68 << 10)      # keeps every pair of bases off the 8K-mod-16K and 2M-mod-4M spacings of r01h_wbench4
        offs[name] = off
        off += nb
    return offs, off


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="packed")
    ap.add_argument("--compact", type=int, default=0)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--launches", type=int, default=12)
    ap.add_argument("--env", default="CartPole-v1")
    args = ap.parse_args()
    import torch
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout(args.env, args.n, seed=0, action_seed=1, reward_f32=bool(args.compact), action_i32=bool(args.compact))
    r.reset(seed=0)
    K, n, O = args.chunk, args.n, r.O
    if args.layout == "sep":
        traj = r.trajectory_buffers(K)
        offs = {k: v.data_ptr() for k, v in traj.items()}
    else:
        import numpy as np

        act_dt = {torch.int64: np.dtype("i8"), torch.int32: np.dtype("i4"), torch.float32: np.dtype("f4")}[r.action_dtype]
        rew_b = 4 if args.compact else 8
        offs, total = carve(None, args.layout, K, n, O, rew_b, act_dt, None, torch)
        with torch.cuda.stream(r.stream):
            block = torch.empty(total + (1 << 30), dtype=torch.uint8, device=r.device)
        base = block.data_ptr()
        shift = (-base) % (1 << 30) if args.layout == "gib" else (-base) % (2 << 20)
        v = lambda name, nb, dt, shape: block[shift + offs[name]: shift + offs[name] + nb].view(dt).view(shape)
        traj = dict(obs=v("obs", K * n * O * 4, torch.float32, (K, n, O)),
                    reward=v("reward", K * n * rew_b, r.reward_dtype, (K, n)),
                    actions=v("actions", K * n * act_dt.itemsize, r.action_dtype, (K, n)),
                    terminated=v("terminated", K * n, torch.uint8, (K, n)),
                    truncated=v("truncated", K * n, torch.uint8, (K, n)))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:          # clock ramp + first touch
        r.rollout_per_step(K, out=traj)
        r.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(r.stream)
        for _ in range(args.launches):
            r.rollout_per_step(K, out=traj)
        e1.record(r.stream)
        r.synchronize()
        res.append(round(e0.elapsed_time(e1) / args.launches / K * 1e3, 3))
    print(json.dumps({"layout": args.layout, "compact": args.compact, "env": args.env, "n": n, "us_per_step": res,
                      "obs_mod_1GiB_MiB": (traj["obs"].data_ptr() % (1 << 30)) >> 20}), flush=True)
    r.close()


if __name__ == "__main__":
    main()
