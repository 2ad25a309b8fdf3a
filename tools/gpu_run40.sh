#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "=== test"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tuned or final_obs" 2>&1 | tail -3
for rep in 1 2 3; do
echo "=== bench (tuned)"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step']*1e3, j['roofline']['frac'], j['config']['placement'], j.get('variants'))"
echo "=== bench (untuned)"; timeout 600 python bench.py --no-cpu-baseline --no-variants --placement-candidates 1 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step']*1e3, j['roofline']['frac'])"
done
} > gpurun_out/run40.log 2>&1
cat gpurun_out/run40.log
