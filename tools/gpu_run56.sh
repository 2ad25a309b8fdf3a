#!/usr/bin/env bash
# TLB warm-up experiment: rollout_kernel touches the output words of step k+W (W = 1, 2, 8) one lane per stream
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do for v in w0 a1 a4; do
  echo -n "$v  "; MXV_LIB_PATH=$PWD/gym_amd/_lib/variants/libmxv_$v.so timeout 300 python bench.py --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); p=j['config']['placement']; print(round(j['value']/1e9,1), 'Ge/s', round(j['ms_per_step']*1e3,3), 'us; tuner best', p['chosen_us_per_step'], 'worst', max(p['us_per_step']+p['mixes_us_per_step']), 'all', sorted(p['us_per_step']))"
done; done
} > gpurun_out/run56.log 2>&1
cat gpurun_out/run56.log
