#!/usr/bin/env python3
"""Kernel micro-benchmark: average step-kernel launch duration per env kind / size / launch mode.

    python tools/kbench.py [--lib PATH/libmxv.so] [--envs CartPole-v1,...] [--n 1048576] [--steps 300]

Prints one JSON line per configuration (HIP-event time over `steps` back-to-back launches / steps).
Used to pick envs-per-lane and launch mode; the .so variants are built by tools/build_variants.sh.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ALGO_B = {"CartPole-v1": 66, "Pendulum-v1": 46, "Acrobot-v1": 74, "MountainCar-v0": 42, "MountainCarContinuous-v0": 42}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--envs", default="CartPole-v1")
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--modes", default="fused,graph,given,fused-final,fusedf32")
    ap.add_argument("--tag", default="")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--layout", default="auto", help="trajectory_buffers(layout=...): auto / sorted / separate / placed")
    args = ap.parse_args()
    if args.lib:
        os.environ["MXV_LIB_PATH"] = os.path.abspath(args.lib)
    import torch
    from gym_amd.rollout import DeviceRollout

    K = args.chunk
    for env in args.envs.split(","):
        for mode in args.modes.split(","):
            f32 = mode.endswith("f32")
            base = mode[:-3] if f32 else mode
            r = DeviceRollout(env, args.n, seed=0, action_seed=1, reward_f32=f32, action_i32=f32)
            r.reset(seed=0)
            if base == "given":      # one launch per step, caller-provided actions, outputs overwritten
                acts = r.sample_actions().clone()

                def go(k):
                    for _ in range(k):
                        r.step(acts, want_final=False)
            elif base == "tape":     # K steps per launch, actions read from a caller-provided [K][N] tape
                seed_run = r.rollout_per_step(K, mode="fused")
                r.synchronize()
                tape = seed_run["actions"].clone()
                tout = {k: seed_run[k] for k in ("obs", "reward", "terminated", "truncated")}

                def go(k):
                    for _ in range(k // K):
                        r.rollout_tape(tape, out=tout)
            elif base.endswith("-final"):  # outputs of every step overwrite one buffer ("final tensors" mode)
                def go(k, m=base[:-6]):
                    for _ in range(k // K):
                        r.rollout(K, mode=m)
            else:                    # trajectory mode: every step writes its own [k][N] slice
                traj = r.trajectory_buffers(K, layout=args.layout)

                def go(k, m=base):
                    for _ in range(k // K):
                        r.rollout_per_step(K, mode=m, out=traj)
            steps = (args.steps // K) * K
            go(K * 2)
            go(steps)
            r.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                ev0.record(r.stream)
                go(steps)
                ev1.record(r.stream)
                r.synchronize()
                best = min(best, ev0.elapsed_time(ev1) / steps * 1e3)
            print(json.dumps({"tag": args.tag, "layout": args.layout, "placement": getattr(r, "last_placement", None), "env": env, "n": args.n, "mode": mode, "chunk": K,
                              "us_per_step": round(best, 3), "env_steps_per_s": float(f"{args.n / (best * 1e-6):.4g}"),
                              "GBs_at_66B": round(ALGO_B[env] * args.n / (best * 1e-6) / 1e9, 1)}), flush=True)
            r.close()


if __name__ == "__main__":
    main()
