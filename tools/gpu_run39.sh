#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 400 python tools/placement_probe3.py 2>&1 | grep -v amdgpu.ids; echo ---; done > gpurun_out/run39.log 2>&1
cat gpurun_out/run39.log
