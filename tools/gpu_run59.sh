#!/usr/bin/env bash
# Occupancy cap via dynamic LDS padding: do balanced rounds of the persistent rollout kernels pay?
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for pad in 0 9216 12288 20480; do
  echo "== pad $pad"
  for n in 1048576 524288; do
  MXV_ROLLOUT_PAD_LDS=$pad timeout 300 python tools/kbench.py --envs Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0,Acrobot-v1,CartPole-v1 --n $n --modes fused,fused-final --steps 1024 --chunk 128 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(j['n'], j['env'].ljust(26), j['mode'].ljust(12), j['us_per_step'])"
  done
done
} > gpurun_out/run59.log 2>&1
cat gpurun_out/run59.log
