#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do TRIALS=4 timeout 600 python tools/placement_probe2.py 2>&1 | grep -v amdgpu.ids; echo ---; done > gpurun_out/run38.log 2>&1
cat gpurun_out/run38.log
