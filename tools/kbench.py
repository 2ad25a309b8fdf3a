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
    ap.add_argument("--modes", default="graph,eager,given,f32")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    if args.lib:
        os.environ["MXV_LIB_PATH"] = os.path.abspath(args.lib)
    import torch
    from gym_amd.rollout import DeviceRollout

    for env in args.envs.split(","):
        for mode in args.modes.split(","):
            r = DeviceRollout(env, args.n, seed=0, action_seed=1, reward_f32=(mode == "f32"),
                              action_i32=(mode == "f32"))
            r.reset(seed=0)
            graph = mode in ("graph", "f32")
            if mode == "given":
                acts = r.sample_actions().clone()

                def go(k):
                    for _ in range(k):
                        r.step(acts, want_final=False)
            else:
                def go(k):
                    r.rollout(k, use_graph=graph)
            go(100)
            go(args.steps)
            r.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                ev0.record(r.stream)
                go(args.steps)
                ev1.record(r.stream)
                r.synchronize()
                best = min(best, ev0.elapsed_time(ev1) / args.steps * 1e3)
            gbs = ALGO_B[env] * args.n / (best * 1e-6) / 1e9
            print(json.dumps({"tag": args.tag, "env": env, "n": args.n, "mode": mode, "us_per_launch": round(best, 3),
                              "env_steps_per_s": args.n / (best * 1e-6), "algo_GBs": round(gbs, 1),
                              "frac_of_8TBs": round(gbs / 8000, 4)}), flush=True)
            r.close()


if __name__ == "__main__":
    main()
