#!/usr/bin/env python3
"""Device-resident use of the engine: cross-entropy search over linear CartPole policies, one candidate policy per env.

Every env of a 65 536-env vector env carries its own weight vector w; the "policy" is one torch expression on the caller's
stream (action = [obs . w > 0]), `DeviceRollout.step(actions)` runs the vector step on the engine's stream, and
`ready()` orders the outputs back for the caller — no host synchronisation inside the loop.  The fused episode statistics
(`enable_episode_stats`) give every candidate's return at the end of its first episode.

    python examples/linear_policy_search.py [--envs 65536] [--iterations 4]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def search(num_envs: int = 1 << 16, iterations: int = 4, horizon: int = 500, seed: int = 0, verbose: bool = True):
    import torch

    from gym_amd.rollout import DeviceRollout

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(seed)
    mean, std = torch.zeros(4, device=dev), torch.ones(4, device=dev)
    env = DeviceRollout("CartPole-v1", num_envs, seed=seed, action_seed=seed + 1)
    env.enable_episode_stats()
    history = []
    for it in range(iterations):
        w = mean + std * torch.randn((num_envs, 4), generator=g, device=dev)          # one candidate per env
        env.reset(seed=seed + it)
        env.ready()
        obs = env.obs
        first_return = torch.full((num_envs,), -1.0, device=dev)                        # return of each env's FIRST episode
        for _ in range(horizon):
            actions = ((obs * w).sum(1) > 0).to(torch.int64)
            obs, _, term, trunc = env.step(actions, want_final=False)
            env.ready()
            done = (term | trunc).bool() & (first_return < 0)
            first_return = torch.where(done, env.ep_return, first_return)
        first_return = torch.where(first_return < 0, torch.full_like(first_return, float(horizon)), first_return)
        elite = first_return.topk(max(8, num_envs // 50)).indices
        mean, std = w[elite].mean(0), w[elite].std(0) + 1e-3
        history.append((float(first_return.mean()), float(first_return[elite].mean())))
        if verbose:
            print(f"iteration {it}: mean return {history[-1][0]:7.1f}   elite mean {history[-1][1]:7.1f}   w = {mean.tolist()}")
    env.close()
    return history, mean.cpu().numpy()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 16)
    ap.add_argument("--iterations", type=int, default=4)
    a = ap.parse_args()
    search(a.envs, a.iterations)
