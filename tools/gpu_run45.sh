#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/config_bench.py 2>&1 | grep -v amdgpu.ids | head -5 > gpurun_out/run45.log
cat gpurun_out/run45.log | cut -c1-260
