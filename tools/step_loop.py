#!/usr/bin/env python3
"""step(actions) at 2^20 CartPole envs: the loop (benchmarks.loops.measure_step_loop) and the kernel, one JSON line.
    python tools/step_loop.py [--compact] [--steps 600] [--env-id CartPole-v1] [--envs 1048576] [--tag x]
Run under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` by tools/gpu_step_traffic.sh for the bytes a launch really moves."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--compact", action="store_true")
ap.add_argument("--obs-state", action="store_true", help="DeviceRollout(obs_carries_state=True): mxv_adopt_obs")
ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--envs", type=int, default=1 << 20)
ap.add_argument("--env-id", default="CartPole-v1")
ap.add_argument("--tag", default="")
ap.add_argument("--plain", action="store_true", help="a bare loop of `steps` launches (for the counter passes: no warm-up spin)")
a = ap.parse_args()

import torch  # noqa: E402

import benchmarks.common as common  # noqa: E402

common.ENV_ID = a.env_id
import benchmarks.loops as loops  # noqa: E402

loops.ENV_ID = a.env_id
if a.plain:
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout(a.env_id, a.envs, seed=0, action_seed=1, reward_f32=a.compact, action_i32=a.compact, obs_carries_state=a.obs_state)
    r.reset(seed=0)
    with torch.cuda.stream(r.stream):
        act = r.sample_actions().clone()
        for _ in range(a.steps):
            r.step(act, want_final=False)
    r.synchronize()
    info = r.handle.last_launch() if hasattr(r.handle, "last_launch") else None
    print(json.dumps({"tag": a.tag, "plain_steps": a.steps, "compact": a.compact, "launch": str(info)}))
    r.close()
else:
    out = loops.measure_step_loop(torch, a.envs, steps=a.steps, compact=a.compact, obs_carries_state=a.obs_state)
    k = loops.measure_step_kernel(torch, a.envs, compact=a.compact)
    print(json.dumps({"tag": a.tag, "env": a.env_id, "envs": a.envs, "compact": a.compact, "obs_state": a.obs_state, "elapsed32": bool(os.environ.get("MXV_ELAPSED32")),
                      "step_loop_us": round(out["us_per_step"], 3), "frac_66B": round(out["roofline"]["frac"], 4),
                      "kernel_us_median": round(k["us_per_launch_median"], 3), "kernel_us_min": round(k["us_per_launch_min"], 3)}))
