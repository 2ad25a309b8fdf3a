#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest vector env"; timeout 900 python -m pytest tests/test_gpu_vector_env.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python - <<'PY'
import sys, time, json
sys.path.insert(0, '.')
from tools.config_bench import compat_loop
for kw in ({}, dict(copy=False), dict(zero_copy=True)):
    for n in (8, 1 << 16, 1 << 20):
        sps, us = compat_loop("CartPole-v1", n, 200 if n < (1 << 20) else 30, **kw)
        print(json.dumps({"kw": kw, "n": n, "us_per_step": round(us, 1), "env_steps_per_s": float(f"{sps:.4g}")}))
PY
} > gpurun_out/run19.log 2>&1
tail -c 3000 gpurun_out/run19.log
