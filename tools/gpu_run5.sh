#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for c in 32 64 100; do timeout 300 python tools/kbench.py --envs CartPole-v1 --steps 1600 --chunk $c --modes fused,fused-final,fusedf32 --tag chunk$c 2>&1 | grep -v amdgpu.ids; done
for v in base px1; do timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs CartPole-v1 --steps 1600 --chunk 64 --modes fused,fused-final 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/run5.log 2>&1
cat gpurun_out/run5.log
