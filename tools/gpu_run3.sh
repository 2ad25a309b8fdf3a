#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
for v in C0E4 C0E2 C0E1 C1E4 C1E2 C0E2w5; do echo "=== variant $v"; timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs CartPole-v1,Pendulum-v1,MountainCar-v0 --steps 512 --modes fused,graph,given 2>&1 | grep -v amdgpu.ids; done
for v in C0E1 C0E2w5; do timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs Acrobot-v1 --n 524288 --steps 256 --modes fused,graph 2>&1 | grep -v amdgpu.ids; done
echo "=== bench fused"; timeout 600 python bench.py --steps 2000 --warmup 200 --chunk 64 --no-cpu-baseline 2>&1 | tail -1
} > gpurun_out/run3.log 2>&1
tail -c 7000 gpurun_out/run3.log
