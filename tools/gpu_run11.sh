#!/usr/bin/env bash
# chunk-size A/B of the default bench on one box (same clocks): is chunk 256 slower than chunk 100?
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
for c in 64 100 128 256 512; do
  echo "=== chunk $c"; timeout 300 python bench.py --no-cpu-baseline --chunk $c --steps $((c*40)) --warmup $((c*4)) 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['config']['chunk'], j['value'], j['ms_per_step']*1e3, j['roofline']['avg_launch_us'], j['roofline']['frac'])"
done
done
} > gpurun_out/run11.log 2>&1
cat gpurun_out/run11.log
