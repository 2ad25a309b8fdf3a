#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== config bench"; timeout 900 python tools/config_bench.py 2>&1 | grep -v amdgpu.ids
echo "=== tab bench"; timeout 300 python tools/tab_bench.py 2>&1 | grep -v amdgpu.ids
echo "=== norm bench"; timeout 300 python tools/norm_bench.py --chunk 128 2>&1 | tail -1
echo "=== bench"; timeout 600 python bench.py 2>&1 | tail -1
} > gpurun_out/run29.log 2>&1
cat gpurun_out/run29.log | cut -c1-420
