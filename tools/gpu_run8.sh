#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for a in "--steps 2000 --warmup 200" "--steps 20000 --warmup 2000" "--steps 100000 --warmup 2000" "--steps 20000 --warmup 2000"; do
echo "=== bench $a"; timeout 600 python bench.py --no-cpu-baseline $a 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_us'])"
done
} > gpurun_out/run8.log 2>&1
cat gpurun_out/run8.log
