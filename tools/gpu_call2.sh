#!/usr/bin/env bash
# GPU call 2 of round 2: RNG contract v2 + rollout_kernel_v3: parity, then same-box A/B against the round-1 library.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
V=gym_amd/_lib/variants
kb() { timeout 200 python tools/kbench.py --lib $V/libmxv_$1.so --tag $1 --envs $2 --n $3 --steps $4 --chunk 256 --modes $5 2>/dev/null | grep '^{' >> $O/ab.jsonl; }
for rep in 1 2; do
  for v in r01 v3 p4 p16 occ5; do kb $v CartPole-v1 1048576 4096 fused,fused-final,fusedf32; done
  for v in r01 v3 impl2; do kb $v Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 1048576 2048 fused,fused-final; kb $v Acrobot-v1 524288 1024 fused; done
done
for n in 32768 65536 131072 262144 524288; do
  for v in r01 v3 nosmall; do kb $v CartPole-v1 $n 4096 fused,fused-final; done
  for v in r01 v3; do kb $v Acrobot-v1,Pendulum-v1,MountainCar-v0 $n 1024 fused; done
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
echo done > $O/finished
