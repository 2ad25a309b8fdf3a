#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40
echo "=== bench fused"; timeout 600 python bench.py --steps 2000 --warmup 200 --chunk 64 2>&1 | tail -2
echo "=== bench graph"; timeout 600 python bench.py --steps 1000 --warmup 128 --chunk 64 --mode graph --no-cpu-baseline 2>&1 | tail -2
echo "=== kbench default"; timeout 600 python tools/kbench.py --envs CartPole-v1,Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 --steps 512 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/kbench.py --envs Acrobot-v1 --n 524288 --steps 256 --modes fused,graph,fusedf32 2>&1 | grep -v amdgpu.ids
for v in C0E4 C1E4 C0E2 C1E2 C0E1 C1E1; do echo "=== variant $v"; timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs CartPole-v1,Pendulum-v1 --steps 512 --modes fused,graph,fusedf32 2>&1 | grep -v amdgpu.ids; done
for v in C0E1 C1E1 C1E2; do timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs Acrobot-v1 --n 524288 --steps 256 --modes fused 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/run2.log 2>&1
tail -c 9000 gpurun_out/run2.log
