#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest gpu (all)"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "=== kbench acrobot"; timeout 300 python tools/kbench.py --envs Acrobot-v1 --n 524288 --modes fused,graph --steps 400 --chunk 100 2>&1 | tail -4
timeout 300 python tools/kbench.py --envs Acrobot-v1 --n 4194304 --modes fused --steps 200 --chunk 100 2>&1 | tail -2
} > gpurun_out/run14.log 2>&1
tail -c 4000 gpurun_out/run14.log
