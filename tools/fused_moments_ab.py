#!/usr/bin/env python3
"""rollout + NormalizeObservation per 2^20-env CartPole step, the batch moments formed (a) by the stand-alone pass that reads the
observations back, (b) by the rollout itself (mxv_set_obs_partials): event-timed on the engine's stream, alternating, one process."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_amd.rollout import DeviceRollout  # noqa: E402

n, K = 1 << 20, 128
for env_id, nn in [(e, n if e == "CartPole-v1" else n >> 1) for e in (sys.argv[1:] or ["CartPole-v1", "MountainCar-v0", "MountainCarContinuous-v0", "Pendulum-v1", "Acrobot-v1"])]:
    r = DeviceRollout(env_id, nn, seed=0, action_seed=1)
    r.reset(seed=0)
    out = r.trajectory_buffers(K, obs_partials=True)
    plain = {k: v for k, v in out.items() if k != "obs_partials"}
    nz = r.make_normalizer()
    y = torch.empty((K, nn, r.O), dtype=torch.float32, device=r.device)

    def timed(fn, reps=5):
        for _ in range(2):
            fn()
        r.stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(r.stream)
        for _ in range(reps):
            fn()
        e1.record(r.stream)
        r.stream.synchronize()
        return e0.elapsed_time(e1) / reps / K * 1e3

    nz.normalize_rewards(out["reward"], out["terminated"], out["truncated"])
    r.fuse_reward_normalizer(nz)
    both = r.trajectory_buffers(K, obs_partials=True, ret_partials=True)
    rets = {k: v for k, v in both.items() if k != "obs_partials"}
    ro = torch.empty_like(both["reward"])
    for rep in range(3):
        res = {"env": env_id, "num_envs": nn,
               "rollout_with_return_partials": timed(lambda: r.rollout_per_step(K, out=rets)),
               "rollout_with_both": timed(lambda: r.rollout_per_step(K, out=both)),
               "normalize_reward": timed(lambda: nz.normalize_rewards(both["reward"], both["terminated"], both["truncated"], out=ro)),
               "normalize_reward_from_partials": timed(lambda: nz.normalize_rewards(both["reward"], both["terminated"], both["truncated"], out=ro, partials=both["ret_partials"])),
               "rollout": timed(lambda: r.rollout_per_step(K, out=plain)),
               "rollout_with_partials": timed(lambda: r.rollout_per_step(K, out=out)),
               "normalize_obs": timed(lambda: nz.normalize_obs(out["obs"], out=y)),
               "normalize_obs_from_partials": timed(lambda: nz.normalize_obs(out["obs"], out=y, partials=out["obs_partials"]))}
        res["pipeline_separate"] = res["rollout"] + res["normalize_obs"]
        res["pipeline_fused"] = res["rollout_with_partials"] + res["normalize_obs_from_partials"]
        res["both_separate"] = res["rollout"] + res["normalize_obs"] + res["normalize_reward"]
        res["both_fused"] = res["rollout_with_both"] + res["normalize_obs_from_partials"] + res["normalize_reward_from_partials"]
        res["reward_only_fused"] = res["rollout_with_return_partials"] + res["normalize_reward_from_partials"]
        res["reward_only_separate"] = res["rollout"] + res["normalize_reward"]
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
    r.close()
    del out, plain, y, both, rets, ro
    torch.cuda.empty_cache()
