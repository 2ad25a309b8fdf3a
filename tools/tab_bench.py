#!/usr/bin/env python3
"""Timing of the tabular toy_text engine: fused K-step rollouts with sampled actions, every step's
obs/actions (int64), reward/prob (float64) and flags written to [K][N] trajectory tensors = 34 B per env-step."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ids", default="FrozenLake-v1,FrozenLake8x8-v1,Taxi-v3,CliffWalking-v0")
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tune", action="store_true", help="placement-tuned trajectory tensors")
    args = ap.parse_args()
    import torch
    from gym_amd.toy_text import TabularRollout

    for gid in args.ids.split(","):
        r = TabularRollout(gid, args.n, seed=0, action_seed=1)
        r.reset(seed=0)
        if args.tune:
            out, rep = r.tuned_trajectory_buffers(args.chunk)
        else:
            out, rep = r.trajectory_buffers(args.chunk), None
        for _ in range(3):
            r.rollout_per_step(args.chunk, out=out)
        r.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(r.stream)
        for _ in range(args.reps):
            r.rollout_per_step(args.chunk, out=out)
        e1.record(r.stream)
        r.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        us = ms * 1e3 / args.chunk
        print(json.dumps({"id": gid, "n": args.n, "chunk": args.chunk, "us_per_step": us,
                          "env_steps_per_s": args.n / (us * 1e-6), "GBs_at_34B": 34 * args.n / (us * 1e-6) / 1e9, "placement": rep}))
        r.close()


if __name__ == "__main__":
    main()
