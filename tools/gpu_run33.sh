#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest blackjack"; timeout 900 python -m pytest tests/test_gpu_blackjack.py -m gpu -x -q 2>&1 | tail -25
timeout 300 python - <<'PY'
import sys, json
sys.path.insert(0, '.')
import torch
from gym_amd import _native
n, K = 1 << 20, 128
dev = torch.device("cuda")
h = _native.Blackjack(n, sab=True, seed=1, action_seed=2)
s = torch.cuda.Stream(); h.set_stream(s.cuda_stream)
with torch.cuda.stream(s):
    obs = torch.zeros((K, 3, n), dtype=torch.int64, device=dev); rew = torch.zeros((K, n), dtype=torch.float64, device=dev)
    term = torch.zeros((K, n), dtype=torch.uint8, device=dev); trunc = torch.zeros((K, n), dtype=torch.uint8, device=dev)
    act = torch.zeros((K, n), dtype=torch.int64, device=dev)
s.synchronize()
h.reset(obs[0])
go = lambda: h.rollout(K, obs, rew, term, trunc, None, act, per_step=True)
for _ in range(3): go()
s.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(10): go()
e1.record(s); s.synchronize()
us = e0.elapsed_time(e1) / 10 / K * 1e3
print(json.dumps({"id": "Blackjack-v1", "n": n, "chunk": K, "us_per_step": us, "env_steps_per_s": n / (us * 1e-6), "GBs_at_42B": 42 * n / (us * 1e-6) / 1e9}))
PY
} > gpurun_out/run33.log 2>&1
tail -c 3500 gpurun_out/run33.log
