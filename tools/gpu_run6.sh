#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/kbench.py --envs CartPole-v1 --steps 1600 --chunk 100 --modes fused,fused-final,fusedf32,graph,given --tag xcd 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/kbench.py --envs Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 --steps 1600 --chunk 100 --modes fused,graph --tag xcd 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/kbench.py --envs Acrobot-v1 --n 524288 --steps 800 --chunk 100 --modes fused,graph --tag xcd 2>&1 | grep -v amdgpu.ids
echo "=== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1
echo "=== bench long"; timeout 600 python bench.py --no-cpu-baseline --steps 20000 --warmup 2000 2>&1 | tail -1
} > gpurun_out/run6.log 2>&1
cat gpurun_out/run6.log
