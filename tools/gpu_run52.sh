#!/usr/bin/env bash
# A/B of MXV_SETTLE_LOADS (entry loads settled before the K-step loop: no vmcnt wait on store acks inside the loop)
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do for v in s0 s1; do
  echo -n "$v  "; MXV_LIB_PATH=$PWD/gym_amd/_lib/variants/libmxv_$v.so timeout 300 python bench.py --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); p=j['config']['placement']; print(round(j['value']/1e9,1), 'Ge/s', round(j['ms_per_step']*1e3,3), 'us; tuner best', p['chosen_us_per_step'], 'worst', max(p['us_per_step']+p['mixes_us_per_step']))"
done; done
for v in s0 s1; do
  echo "== $v tab"; MXV_LIB_PATH=$PWD/gym_amd/_lib/variants/libmxv_$v.so timeout 300 python tools/tab_bench.py --ids Taxi-v3,FrozenLake-v1 --tune 2>&1 | tail -2 | cut -c1-250
  echo "== $v configs"; MXV_LIB_PATH=$PWD/gym_amd/_lib/variants/libmxv_$v.so timeout 300 python tools/config_bench.py 2>&1 | tail -8 | cut -c1-250
done
} > gpurun_out/run52.log 2>&1
cat gpurun_out/run52.log
