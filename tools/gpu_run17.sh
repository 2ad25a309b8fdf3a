#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest toytext"; timeout 900 python -m pytest tests/test_gpu_toytext.py -m gpu -x -q 2>&1 | tail -30
echo "=== tab bench"; timeout 300 python tools/tab_bench.py 2>&1 | tail -4
} > gpurun_out/run17.log 2>&1
tail -c 6000 gpurun_out/run17.log
