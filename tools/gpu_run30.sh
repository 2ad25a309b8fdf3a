#!/usr/bin/env bash
bash tools/gpu_ab.sh "n0 w1 w2 w4" CartPole-v1,Pendulum-v1,MountainCar-v0 fused,fusedf32 1048576 2
cp gpurun_out/ab.log gpurun_out/ab_light.log
timeout 300 python tools/tab_bench.py --ids Taxi-v3 2>&1 | tail -1
