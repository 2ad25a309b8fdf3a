#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest normalize"; timeout 900 python -m pytest tests/test_gpu_normalize.py -m gpu -x -q 2>&1 | tail -5
echo "=== norm bench"; timeout 300 python tools/norm_bench.py --chunk 128 2>&1 | tail -1
} > gpurun_out/run16.log 2>&1
tail -c 3000 gpurun_out/run16.log
