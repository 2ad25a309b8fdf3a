#!/usr/bin/env python3
"""Timing of the §8(f)-2 normalisation kernels on trajectory chunks produced by the fused rollout (one MI355X).
Reports, per launch group, us per vector step and the HBM rate on the bytes each pass must move:
  obs sums : read 4*O B / row            obs apply: read 4*O + write 8*O (float64 out) or 4*O (float32 out)
  rew sums : read 8 + 2 B / env-step     rew apply: read 8 + write 8
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="CartPole-v1")
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    n, K = args.n, args.chunk
    dr = DeviceRollout(args.env, n, seed=0, action_seed=1)
    dr.reset(seed=0)
    tr = dr.rollout_per_step(K)
    dr.synchronize()
    O = dr.O
    x, r, te, tc = tr["obs"], tr["reward"], tr["terminated"], tr["truncated"]
    s = dr.stream
    no = _native.Norm(O, n, stream=s.cuda_stream)
    nr = _native.Norm(1, n, stream=s.cuda_stream)
    with torch.cuda.stream(s):
        y64 = torch.empty((K, n, O), dtype=torch.float64, device="cuda")
        y32 = torch.empty((K, n, O), dtype=torch.float32, device="cuda")
        o64 = torch.empty((K, n), dtype=torch.float64, device="cuda")
        so = torch.empty((K, 2 * O), dtype=torch.float64, device="cuda")
        sr = torch.empty((K, 2), dtype=torch.float64, device="cuda")

    def timed(fn, bytes_per_env_step):
        fn(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(args.reps):
            fn()
        e1.record(s)
        s.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        return {"us_per_step": ms * 1e3 / K, "GBs": bytes_per_env_step * n * K / (ms * 1e-3) / 1e9}

    out = {"env": args.env, "n": n, "chunk": K}
    out["obs_sums"] = timed(lambda: no.obs_sums(K, x, so), 4 * O)
    out["obs_apply_f64"] = timed(lambda: no.obs_apply(K, x, y64, False, 1e-8, so, 1, n), 12 * O)
    out["obs_apply_f32"] = timed(lambda: no.obs_apply(K, x, y32, True, 1e-8, so, 1, n), 8 * O)
    out["obs_total_f64"] = timed(lambda: no.observations(K, x, y64, False, 1e-8), 16 * O)
    out["obs_total_f32"] = timed(lambda: no.observations(K, x, y32, True, 1e-8), 12 * O)
    out["rew_sums"] = timed(lambda: nr.reward_sums(K, r, False, te, tc, 0.99, sr), 10)
    out["rew_apply"] = timed(lambda: nr.reward_apply(K, r, False, o64, 1e-8, sr, 1, n), 16)
    out["rew_total"] = timed(lambda: nr.rewards(K, r, False, te, tc, o64, 0.99, 1e-8), 26)
    out["rollout_fused"] = timed(lambda: dr.rollout_per_step(K, out=tr), 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
