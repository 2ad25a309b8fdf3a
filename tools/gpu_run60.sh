#!/usr/bin/env bash
# rollout_kernel with a 48-B reset entry (7.5 KiB LDS) at MXV_ROLLOUT_MIN_WAVES 1 / 5 / 6 (VGPR caps 128 / 96 / 80)
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2; do for v in l0 l5 l6; do
  echo -n "$v  "; MXV_LIB_PATH=$PWD/gym_amd/_lib/variants/libmxv_$v.so timeout 300 python bench.py --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); p=j['config']['placement']; print(round(j['value']/1e9,1), 'Ge/s', round(j['ms_per_step']*1e3,3), 'us; tuner best', p['chosen_us_per_step'], 'worst', max(p['us_per_step']+p['mixes_us_per_step']))"
done; done
for v in l0 l5 l6; do
  echo "== $v"; timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --envs Acrobot-v1 --n 524288 --modes fused,fused-final --steps 1024 --chunk 128 2>&1 | grep "^{" | cut -c1-170
  timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --envs CartPole-v1 --n 1048576 --modes fused-final --steps 1024 --chunk 128 2>&1 | grep "^{" | cut -c1-170
done
} > gpurun_out/run60.log 2>&1
cat gpurun_out/run60.log
