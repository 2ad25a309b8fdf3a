#!/usr/bin/env bash
# GPU call 1 of round 2: new parity tests, bench with the driver's arguments, shard-size sweep, single-allocation placement.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
# placement: 5 fresh processes per layout
for rep in 1 2 3 4 5; do for lay in sep packed mib2 stagger gib; do
  timeout 120 python tools/placement_block.py --layout $lay 2>/dev/null | grep '^{' >> $O/placement.jsonl
done; done
for rep in 1 2 3; do for lay in sep packed stagger; do
  timeout 120 python tools/placement_block.py --layout $lay --compact 1 2>/dev/null | grep '^{' >> $O/placement.jsonl
done; done
# shard-size sweep: us/step of the fused trajectory rollout, E = 2 (product) vs E = 1 variant
for n in 32768 65536 131072 262144 524288 1048576; do
  for v in e1 e2; do
    lib=gym_amd/_lib/variants/libmxv_$v.so
    timeout 200 python tools/kbench.py --lib $lib --tag $v --envs CartPole-v1 --n $n --steps 4096 --chunk 256 --modes fused,fused-final 2>/dev/null | grep '^{' >> $O/sweep.jsonl
  done
  timeout 200 python tools/kbench.py --tag prod --envs Acrobot-v1,Pendulum-v1,MountainCar-v0 --n $n --steps 1024 --chunk 256 --modes fused 2>/dev/null | grep '^{' >> $O/sweep.jsonl
done
echo done > $O/finished
