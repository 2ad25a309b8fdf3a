#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest gpu (all)"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
} > gpurun_out/run21.log 2>&1
tail -c 3000 gpurun_out/run21.log
