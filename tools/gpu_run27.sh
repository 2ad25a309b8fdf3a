#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
} > gpurun_out/run27.log 2>&1
cat gpurun_out/run27.log
bash tools/gpu_ab.sh "oldv1 oldv2 new" CartPole-v1,Pendulum-v1,MountainCar-v0 fused,fusedf32 1048576 2
cp gpurun_out/ab.log gpurun_out/ab_light.log
bash tools/gpu_ab.sh "oldv1 new" Acrobot-v1 fused 524288 2
