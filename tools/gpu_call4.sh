#!/usr/bin/env bash
# GPU call 4 of round 2: new tests (mixed single launch, distributions), mixed-batch timing, tile-major layout A/B, placement.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_distributions.py tests/test_gpu_comm.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
V=gym_amd/_lib/variants
kb() { timeout 200 python tools/kbench.py --lib $V/libmxv_$1.so --tag $1 --envs $2 --n $3 --steps $4 --chunk 256 --modes $5 2>/dev/null | grep '^{' >> $O/ab.jsonl; }
for rep in 1 2 3; do
  for v in v3 tilemajor r01; do kb $v CartPole-v1 1048576 4096 fused,fusedf32; done
done
for v in v3 tilemajor; do kb $v Pendulum-v1,MountainCar-v0 1048576 2048 fused; done
for rep in 1 2 3; do for lay in sep packed stagger gib; do
  timeout 120 python tools/placement_block.py --layout $lay 2>/dev/null | grep '^{' >> $O/placement.jsonl
done; done
timeout 300 python tools/config_bench_dist.py --chunk 256 --steps 4096 > $O/configs_dist.jsonl 2> $O/configs_dist.err
MXV_MIXED_MULTI_LAUNCH=1 timeout 300 python tools/config_bench_dist.py --chunk 256 --steps 4096 >> $O/configs_dist.jsonl 2>> $O/configs_dist.err
echo done > $O/finished
