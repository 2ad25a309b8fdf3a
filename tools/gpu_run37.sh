#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python tools/placement_probe.py 2>&1 | grep -v amdgpu.ids; echo ---; done > gpurun_out/run37.log 2>&1
cat gpurun_out/run37.log
