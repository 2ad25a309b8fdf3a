#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python - > gpurun_out/run20.log 2>&1 <<'PY'
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
import gym_amd
n = 1 << 20
for kw in ({}, dict(zero_copy=True)):
    env = gym_amd.make("CartPole-v1", n, **kw)
    env.reset(seed=0)
    env.action_space.seed(0)
    acts = [env.action_space.sample() for _ in range(4)]
    for i in range(3): env.step(acts[i % 4])
    pr = cProfile.Profile(); pr.enable()
    for i in range(20): env.step(acts[i % 4])
    pr.disable()
    print("====", kw)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    env.close()
PY
grep -v "^$" gpurun_out/run20.log | head -70
