#!/usr/bin/env python3
"""Timing of the Blackjack-v1 engine: fused K-step rollouts with sampled actions, every step's observation (three
int64 columns), reward (float64), flags and actions (int64) written to [K][...][N] trajectory tensors = 42 B per env-step."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    import torch
    from gym_amd import _native

    n, K = args.n, args.chunk
    dev = torch.device("cuda", 0)
    h = _native.Blackjack(n, seed=0, action_seed=1)
    obs = torch.empty((K, 3, n), dtype=torch.int64, device=dev)
    rew = torch.empty((K, n), dtype=torch.float64, device=dev)
    term = torch.empty((K, n), dtype=torch.uint8, device=dev)
    trunc = torch.empty((K, n), dtype=torch.uint8, device=dev)
    act = torch.empty((K, n), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    h.reset()
    run = lambda: h.rollout(K, obs, rew, term, trunc, None, actions_out_dev=act, per_step=True)   # noqa: E731
    for _ in range(3):
        run()
    h.sync()
    import time
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    h.sync()
    us = (time.perf_counter() - t0) / args.reps / K * 1e6
    ended = float(((term | trunc) != 0).float().mean().item())
    print(json.dumps({"tag": args.tag, "id": "Blackjack-v1", "n": n, "chunk": K, "us_per_step": us, "env_steps_per_s": n / (us * 1e-6),
                      "GBs_at_42B": 42 * n / (us * 1e-6) / 1e9, "episodes_ended_per_env_step": ended,
                      "lib": os.environ.get("MXV_LIB_PATH", "default")}))
    h.close()


if __name__ == "__main__":
    main()
