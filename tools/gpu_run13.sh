#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== pytest normalize"; timeout 900 python -m pytest tests/test_gpu_normalize.py -m gpu -x -q 2>&1 | tail -30
echo "=== norm bench"; for c in 32 128; do timeout 300 python tools/norm_bench.py --chunk $c 2>&1 | tail -1; done
timeout 300 python tools/norm_bench.py --env Acrobot-v1 --n 524288 --chunk 64 2>&1 | tail -1
timeout 300 python tools/norm_bench.py --env Pendulum-v1 --n 524288 --chunk 64 2>&1 | tail -1
} > gpurun_out/run13.log 2>&1
tail -c 6000 gpurun_out/run13.log
