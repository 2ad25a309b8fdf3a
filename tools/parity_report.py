#!/usr/bin/env python3
"""Measured parity of the HIP engine against the REFERENCE's own outputs (tests/golden/*_p1.npz single steps from hand-set states,
*_p2_{default,short}.npz teacher-forced SyncVectorEnv trajectories; made by running openai/gym 0.26.2, tests/golden/make_golden.py): per
env kind the number of mask / elapsed mismatches, the largest observation distance in float32 ulps, the largest relative / absolute
reward difference, the largest relative fp64 state difference — the numbers behind the README's "Parity" table.  Needs the GPU.

    python tools/parity_report.py > gpurun_out/parity_report.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import ENV_NAMES, HipEngine, load_golden, ulps32  # noqa: E402


def fold(acc, obs, robs, rew, rrew, term, rterm, trunc, rtrunc, st, rst, nd):
    acc["steps"] += int(rterm.size)
    acc["mask_mismatches"] += int((term != rterm).sum() + (trunc != rtrunc).sum())
    u = ulps32(obs[nd], robs[nd])
    acc["obs_max_ulps"] = max(acc["obs_max_ulps"], int(u.max(initial=0)))
    acc["obs_elements_off_by_one_or_more"] += int((u > 0).sum())
    acc["obs_elements"] += int(u.size)
    d = np.abs(rew - rrew)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(rrew != 0, d / np.abs(rrew), 0.0)
    acc["reward_max_abs"] = max(acc["reward_max_abs"], float(d.max(initial=0)))
    acc["reward_max_rel"] = max(acc["reward_max_rel"], float(rel.max(initial=0)))
    acc["rewards_not_bit_equal"] += int((rew != rrew).sum())
    if st is not None:
        with np.errstate(divide="ignore", invalid="ignore"):
            srel = np.where(rst[nd] != 0, np.abs(st[nd] - rst[nd]) / np.abs(rst[nd]), 0.0)
        acc["state_max_rel"] = max(acc["state_max_rel"], float(srel.max(initial=0)))


def main():
    out = {}
    for name in ENV_NAMES:
        acc = dict(steps=0, mask_mismatches=0, obs_max_ulps=0, obs_elements_off_by_one_or_more=0, obs_elements=0, reward_max_abs=0.0,
                   reward_max_rel=0.0, rewards_not_bit_equal=0, state_max_rel=0.0)
        g = load_golden(name, "p1")
        n = len(g["action"])
        eng = HipEngine(name, n, 0, autoreset=False)
        eng.set_state(g["state0"].T, np.where(g["fresh"] == 1, 0, 5).astype(np.int32))     # (MountainCarContinuous: float32 state after any step)
        obs, rew, term, trunc, _ = eng.step(g["action"])
        st, _ = eng.get_state()
        every = np.ones(n, bool)
        fold(acc, obs, g["obs"], rew, g["reward"], term, g["terminated"].astype(bool), trunc, np.zeros(n, bool), st.T, g["state1"], every)
        for tag in ("default", "short"):
            g = load_golden(name, f"p2_{tag}")
            T, N = g["action"].shape
            eng = HipEngine(name, N, int(g["max_episode_steps"]), autoreset=True)
            for t in range(T):
                eng.set_state(g["state_pre"][t].T, g["elapsed_pre"][t])
                obs, rew, term, trunc, fin = eng.step(g["action"][t])
                st, _ = eng.get_state()
                done = (g["terminated"][t] | g["truncated"][t]).astype(bool)
                fold(acc, obs, g["obs"][t], rew, g["reward"][t], term, g["terminated"][t].astype(bool), trunc, g["truncated"][t].astype(bool),
                     st.T, g["state_post"][t], ~done)
                if done.any():
                    acc["obs_max_ulps"] = max(acc["obs_max_ulps"], int(ulps32(fin[done], g["final_obs"][t][done]).max(initial=0)))
        acc["obs_fraction_not_bit_equal"] = acc.pop("obs_elements_off_by_one_or_more") / max(acc["obs_elements"], 1)
        out[name] = acc
    print(json.dumps({"what": "HIP engine vs the reference's own outputs (goldens P1 + P2 default/short)", "kinds": out}, indent=1))


if __name__ == "__main__":
    main()
