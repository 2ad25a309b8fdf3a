#!/usr/bin/env bash
# GPU call 3 of round 2: full GPU suite on the v3 kernel + comm tests, rocprofv3 stats + PMC traffic of the bench command.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
bash tools/gpu_profile.sh r02c fused 256 > $O/profile.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
echo done > $O/finished
