#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/run53_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/run53_tests.log
tail -3 gpurun_out/run53_tests.log
bash tools/gpu_profile.sh r01j fused 256 > gpurun_out/run53_prof.log 2>&1
tail -3 gpurun_out/run53_prof.log
timeout 300 python bench.py > gpurun_out/bench_r01j.json 2> gpurun_out/bench_r01j.err; tail -1 gpurun_out/bench_r01j.json | cut -c1-600
