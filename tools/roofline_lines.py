#!/usr/bin/env python3
"""Bench-format lines with BOTH rooflines (SURVEY.md §8d "report both") for the fused rollout of every env kind, from one GPU
call's evidence: tools/kbench.py timings (times.jsonl) and rocprofv3 --pmc passes of the same command (SQ_INSTS_VALU /
FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) under gpurun_out/<tag>/.

    python tools/roofline_lines.py r02e > profiles/r02e_rooflines.jsonl

roofline.bound is the tighter of
  hbm  : algorithmic bytes per env-step (4*O + 4 + 4 + 2 + 16*S/K) x env-steps per launch / launch time, peak 8 TB/s
  valu : VALU instructions issued per launch (PMC, summed over waves) / launch time, peak = 1024 SIMDs x 2.4 GHz / 4 cycles
         per wave64 instruction = 6.144e11 wave-instructions/s (fp64 and fp32/int VALU ops issue at the same rate on CDNA4)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02e"
src = os.path.join(ROOT, "gpurun_out", tag)
DIMS = {"CartPole-v1": (4, 4), "Pendulum-v1": (2, 3), "Acrobot-v1": (4, 6), "MountainCar-v0": (2, 2), "MountainCarContinuous-v0": (2, 2)}
VALU_PEAK = 1024 * 2.4e9 / 4
HBM_PEAK = 8000.0
K = 256


def counters(prefix, env):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, f"{prefix}_{env}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


times = {}
for line in open(os.path.join(src, "times.jsonl")):
    j = json.loads(line)
    times[(j["env"], j["mode"])] = j
for env, (S, O) in DIMS.items():
    t = times.get((env, "fused"))
    if not t:
        continue
    n, us = t["n"], t["us_per_step"]
    c = {**counters("pmc", env), **counters("pmcf", env), **counters("pmcw", env)}
    launch_s = us * 1e-6 * K
    algo_b = 4 * O + 4 + 4 + 2 + 16.0 * S / K
    hbm = algo_b * n * K / launch_s / 1e9
    valu = c.get("SQ_INSTS_VALU", 0.0) / launch_s
    traffic = c.get("FETCH_SIZE", 0.0) * 1024 / 0.5 + c.get("WRITE_SIZE", 0.0) * 1024   # gfx950: FETCH_SIZE counts half the bytes
    waves = c.get("SQ_WAVES", 0.0)
    bound = "valu" if valu / VALU_PEAK > hbm / HBM_PEAK else "hbm"
    line = {
        "metric": f"env-steps/sec, {env}, num_envs={n}, fused rollout (K={K}), 1 MI355X", "value": n / (us * 1e-6), "unit": "env-steps/s",
        "ms_per_step": us * 1e-3, "dtype": "f64", "config": {"workload": f"{env}, {n} envs, on-device autoreset + Philox actions, "
                                                             "trajectory tensors [K][N]", "chunk": K},
        "roofline": {"bound": bound,
                     "hbm": {"achieved": hbm, "peak": HBM_PEAK, "unit": "GB/s", "frac": hbm / HBM_PEAK,
                             "algorithmic_bytes_per_env_step": algo_b, "traffic": traffic,
                             "traffic_bytes_per_env_step": traffic / (n * K)},
                     "valu": {"achieved": valu / 1e9, "peak": VALU_PEAK / 1e9, "unit": "G wave-instructions/s", "frac": valu / VALU_PEAK,
                              "valu_instructions_per_wave_step": c.get("SQ_INSTS_VALU", 0.0) / max(waves, 1) / K,
                              "salu_instructions_per_wave_step": c.get("SQ_INSTS_SALU", 0.0) / max(waves, 1) / K,
                              "envs_per_lane": int(round(n / max(waves, 1) / 64))},
                     "frac": max(valu / VALU_PEAK, hbm / HBM_PEAK)},
        "cache_resident_us_per_step": times.get((env, "fused-final"), {}).get("us_per_step"),
    }
    print(json.dumps(line))
