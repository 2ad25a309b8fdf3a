#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_profile.sh r01i fused 256 > gpurun_out/run32.log 2>&1
tail -3 gpurun_out/run32.log
