#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01h.json
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
} > gpurun_out/run22.log 2>&1
tail -c 4000 gpurun_out/run22.log
