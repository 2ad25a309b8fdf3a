#!/usr/bin/env bash
# GPU call 6 of round 2: medium-range sincos (mx_sincos) — full parity suite, then same-box A/B against the ocml build.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
V=gym_amd/_lib/variants
kb() { timeout 200 python tools/kbench.py --lib $V/libmxv_$1.so --tag $1 --envs $2 --n $3 --steps $4 --chunk 256 --modes $5 2>/dev/null | grep '^{' >> $O/ab.jsonl; }
for rep in 1 2; do
  for v in ocmltrig v3; do
    kb $v Acrobot-v1 524288 1024 fused,fused-final
    kb $v Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 1048576 2048 fused,fused-final
    kb $v Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 524288 2048 fused
    kb $v Acrobot-v1,Pendulum-v1,MountainCar-v0 32768 2048 fused
  done
done
timeout 300 python tools/config_bench_dist.py --chunk 256 --steps 4096 >> $O/configs_dist.jsonl 2>> $O/configs_dist.err
echo done > $O/finished
