#!/usr/bin/env bash
bash tools/gpu_ab.sh "n0 n1" CartPole-v1,Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 fused,fusedf32 1048576 3
cp gpurun_out/ab.log gpurun_out/ab_light.log
bash tools/gpu_ab.sh "n0 n1" Acrobot-v1 fused 524288 2
