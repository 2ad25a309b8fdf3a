#!/usr/bin/env bash
# after the settle fix: re-check envs-per-lane and rollout_kernel_v2 for CartPole; new wrapper test + full suite
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/run54_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/run54_tests.log
tail -3 gpurun_out/run54_tests.log
{
for rep in 1 2; do for v in base e1 e3 v2all v2w2; do
  echo -n "$v  "; MXV_LIB_PATH=$PWD/gym_amd/_lib/variants/libmxv_$v.so timeout 300 python bench.py --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); p=j['config']['placement']; print(round(j['value']/1e9,1), 'Ge/s', round(j['ms_per_step']*1e3,3), 'us; tuner best', p['chosen_us_per_step'], 'worst', max(p['us_per_step']+p['mixes_us_per_step']))"
done; done
} > gpurun_out/run54.log 2>&1
cat gpurun_out/run54.log
