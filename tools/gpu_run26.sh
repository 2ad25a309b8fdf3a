#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
} > gpurun_out/run26.log 2>&1
cat gpurun_out/run26.log
bash tools/gpu_ab.sh "v1 v2" CartPole-v1,Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 fused,fused-final 1048576 2
bash tools/gpu_ab.sh "v1 v2" Acrobot-v1 fused 524288 2
