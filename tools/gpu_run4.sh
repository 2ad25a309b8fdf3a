#!/usr/bin/env bash
# Round-1 re-entry check: full GPU parity suite, default bench line, rocprof evidence for the fused launch shape.
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "=== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25
echo "=== bench default"; timeout 600 python bench.py 2>&1 | tail -1
echo "=== profile fused"; bash tools/gpu_profile.sh r01f fused 100 2>&1 | tail -5
} > gpurun_out/run4.log 2>&1
tail -c 5000 gpurun_out/run4.log
