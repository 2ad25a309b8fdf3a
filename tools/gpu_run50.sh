#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "Acrobot or acrobot or known" 2>&1 | tail -3
timeout 300 python tools/kbench.py --envs Acrobot-v1 --n 524288 --modes fused --steps 800 --chunk 100 2>&1 | tail -1
