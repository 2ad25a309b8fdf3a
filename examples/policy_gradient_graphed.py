#!/usr/bin/env python3
"""A learner in the loop at a PPO-sized batch, with the sampling loop recorded ONCE in a hipGraph.

REINFORCE with a linear-logistic policy on CartPole-v1, 4 096 envs x 64 steps per iteration.  At this batch size a vector step is 3 us
of kernel and the loop `actions = policy(obs); env.step(actions)` is bound by launches (about 30 us per step from Python); after
`DeviceRollout.graphed_loop(policy, K, on_step=record)` the 64 steps — policy kernels, env steps, the copies into the trajectory
tensors — are one `graph.replay()` (about 12 us per step).  The step index of the engine's Philox streams lives in device memory
(mxv_set_device_clock), so every replay continues the streams; the policy's own randomness is torch's graph-safe CUDA generator.
The weights are a static tensor the update writes in place, so the recorded graph always reads the current policy.

    python examples/policy_gradient_graphed.py [--envs 4096] [--iterations 30] [--eager]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def train(num_envs: int = 4096, iterations: int = 30, K: int = 64, lr: float = 10.0, gamma: float = 0.99, seed: int = 0, graphed: bool = True,
          verbose: bool = True):
    import torch

    from gym_amd.rollout import DeviceRollout

    torch.manual_seed(seed)
    env = DeviceRollout("CartPole-v1", num_envs, seed=seed, action_seed=seed + 1)
    env.enable_episode_stats()
    env.reset(seed=seed)
    dev = env.device
    w = torch.zeros(4, device=dev)                                            # static: updated in place
    traj = {"obs": torch.empty((K, num_envs, 4), device=dev), "act": torch.empty((K, num_envs), device=dev),
            "rew": torch.empty((K, num_envs), device=dev), "done": torch.empty((K, num_envs), device=dev),
            "ep_len": torch.zeros((K, num_envs), device=dev)}

    def policy(obs):
        return (torch.rand(num_envs, device=dev) < torch.sigmoid(obs @ w)).to(torch.int64)

    pending = {}

    def policy_recording(obs):
        pending["obs"] = obs.clone()                                          # the observation the action is chosen on
        pending["act"] = policy(obs)
        return pending["act"]

    def record(k):
        traj["obs"][k].copy_(pending["obs"])
        traj["act"][k].copy_(pending["act"])
        traj["rew"][k].copy_(env.reward)
        done = torch.bitwise_or(env.terminated, env.truncated)
        traj["done"][k].copy_(done)
        traj["ep_len"][k].copy_(env.ep_length * done)                         # length of the episode that ended here (fused statistics)

    if graphed:
        graph = env.graphed_loop(policy_recording, K, on_step=record)
        sample = graph.replay
    else:
        def sample():
            with torch.cuda.stream(env.stream):
                for k in range(K):
                    env.step(policy_recording(env.obs), want_final=False)
                    record(k)

    history = []
    with torch.cuda.stream(env.stream):
        for it in range(iterations):
            t0 = time.perf_counter()
            sample()
            env.stream.synchronize()
            dt = time.perf_counter() - t0
            # returns-to-go inside the chunk (episodes cut at the chunk's end bootstrap with 0: fine for a demonstration)
            ret = torch.zeros(num_envs, device=dev)
            G = torch.empty_like(traj["rew"])
            for k in range(K - 1, -1, -1):
                ret = traj["rew"][k] + gamma * ret * (1.0 - traj["done"][k])
                G[k] = ret
            adv = (G - G.mean()) / (G.std() + 1e-8)
            p = torch.sigmoid(traj["obs"] @ w)
            grad = ((adv * (traj["act"] - p)).unsqueeze(-1) * traj["obs"]).mean(dim=(0, 1))
            w.add_(lr * grad)
            ended = traj["done"].sum().clamp(min=1.0)
            mean_len = float(traj["ep_len"].sum() / ended)
            history.append({"iteration": it, "mean_episode_length": mean_len, "episodes": int(ended), "us_per_step": dt / K * 1e6})
            if verbose:
                print(f"iteration {it:3d}: {int(ended):6d} episodes ended, mean length {mean_len:7.1f}, sampling {dt / K * 1e6:6.1f} us per vector step")
    env.close()
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--eager", action="store_true", help="one call per step instead of the recorded graph")
    a = ap.parse_args()
    h = train(a.envs, a.iterations, graphed=not a.eager)
    print(f"mean episode length {h[0]['mean_episode_length']:.1f} -> {h[-1]['mean_episode_length']:.1f}; "
          f"sampling {sum(x['us_per_step'] for x in h[2:]) / max(1, len(h) - 2):.1f} us per vector step")
