#!/usr/bin/env bash
# Round-end evidence: full GPU suite, smoke, rocprofv3 stats + PMC of the bench command, bench with the driver's arguments.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${1:-r02n}
O=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
bash tools/gpu_profile.sh $R fused 256 > $O/profile.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo done > $O/finished
