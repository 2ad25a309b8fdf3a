#!/usr/bin/env bash
# The round-end evidence call (rounds 5 and 6): full GPU suite + smoke, rocprofv3 kernel stats and PMC traffic of the bench command (headline,
# its compact twin, and one pass over everything the line's variants launch), bench with the driver's arguments and with the defaults, the
# parity report, Blackjack's VALU counters.      gpurun --timeout 3000 -- 'bash tools/gpu_call.sh r6'   ->  gpurun_out/r6/ -> profiles/r6/r6_*
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${1:-r6}
O=$GRAFT_REPO_ROOT/gpurun_out/$R; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
bash tools/gpu_profile.sh $R fused 256 > $O/profile.log 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/variants_trace -o bench -- $B > $O/variants_trace.log 2>&1
C="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --compact-outputs --steps 1024 --warmup 256"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/compact_pmc_$c -o bench -- $C > $O/compact_pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
find $O/variants_trace $O/compact_pmc_FETCH_SIZE $O/compact_pmc_WRITE_SIZE -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --variants-file $O/bench_driver_args_variants.json --headline-file $O/bench_driver_args_headline.json > $O/bench_driver_args.json 2> $O/bench_driver_args.err ) 2> $O/bench_driver_args.time
timeout 400 python bench.py --variants-file $O/bench_default_variants.json --headline-file $O/bench_default_headline.json > $O/bench_default.json 2> $O/bench_default.err
python tools/parity_report.py > $O/parity_report.json 2> /dev/null
bash tools/gpu_valu_bj.sh $R > $O/valu_bj.log 2>&1
echo done > $O/finished
