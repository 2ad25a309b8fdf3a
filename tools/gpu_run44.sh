#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 300 python tools/tab_bench.py --ids Taxi-v3,FrozenLake-v1 2>&1 | grep -v amdgpu | cut -c1-200
timeout 300 python tools/tab_bench.py --ids Taxi-v3,FrozenLake-v1 --tune 2>&1 | grep -v amdgpu | cut -c1-700
} > gpurun_out/run44.log 2>&1
cat gpurun_out/run44.log
