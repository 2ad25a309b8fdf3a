#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do
for sp in 0 150 1000; do
  echo -n "spinup=$sp  "; timeout 300 python bench.py --no-cpu-baseline --no-variants --spinup-ms $sp 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step']*1e3)"
done
echo -n "steps=204800 (1.4 s timed) "; timeout 300 python bench.py --no-cpu-baseline --no-variants --steps 204800 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step']*1e3)"
done
} > gpurun_out/run36.log 2>&1
cat gpurun_out/run36.log
