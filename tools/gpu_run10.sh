#!/usr/bin/env bash
# Re-entry verification of HEAD on one MI355X: smoke, the GPU parity suite, the default bench line, then the
# rocprofv3 kernel-trace + FETCH/WRITE PMC passes of the default bench shape (fused, chunk 256).
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "=== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "=== bench"; timeout 600 python bench.py 2>&1 | tail -2 | tee gpurun_out/bench_r01g.json
} > gpurun_out/run10.log 2>&1
bash tools/gpu_profile.sh r01g fused 256 >> gpurun_out/run10.log 2>&1
tail -c 5000 gpurun_out/run10.log
