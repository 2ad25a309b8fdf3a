#!/usr/bin/env bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vector_env.py -m gpu -q -x 2>&1 | tail -8
bash tools/gpu_ab.sh "pE1 pE2 qE1 qE2" CartPole-v1,Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 fused 1048576 1
cp gpurun_out/ab.log gpurun_out/ab_light.log
bash tools/gpu_ab.sh "pE1 qE1 qE2" Acrobot-v1 fused 524288 1
