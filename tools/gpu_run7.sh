#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vector_env.py -m gpu -q -x 2>&1 | tail -12
} > gpurun_out/run7.log 2>&1
cat gpurun_out/run7.log
bash tools/gpu_ab.sh "old new" CartPole-v1 fused,fused-final,graph 1048576 2
