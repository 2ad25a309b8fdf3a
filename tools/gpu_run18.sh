#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "=== config bench"; timeout 600 python tools/config_bench.py 2>&1 | tail -12
echo "=== compat loop sizes"; timeout 300 python - <<'PY'
import sys, time, json
sys.path.insert(0, '.')
from tools.config_bench import compat_loop
for n in (8, 64, 1024, 4096, 8192):
    sps, us = compat_loop("CartPole-v1", n, 500)
    print(json.dumps({"n": n, "us_per_step": round(us, 1), "env_steps_per_s": float(f"{sps:.4g}")}))
PY
} > gpurun_out/run18.log 2>&1
tail -c 5000 gpurun_out/run18.log
