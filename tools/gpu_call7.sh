#!/usr/bin/env bash
# GPU call 7 of round 2: occupancy pinned to 4 waves/SIMD (LDS pad) vs not; config 5 both dispatch modes; rooflines of the final kernels.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $O
V=gym_amd/_lib/variants
kb() { timeout 200 python tools/kbench.py --lib $V/libmxv_$1.so --tag $1 --envs $2 --n $3 --steps $4 --chunk 256 --modes $5 2>/dev/null | grep '^{' >> $O/ab.jsonl; }
for rep in 1 2; do
  for v in v3 occ4; do
    kb $v Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0,CartPole-v1 1048576 2048 fused
    kb $v Pendulum-v1,Acrobot-v1,MountainCar-v0 524288 2048 fused
    kb $v Pendulum-v1,CartPole-v1 131072 2048 fused
    kb $v Pendulum-v1,CartPole-v1,MountainCar-v0 32768 2048 fused
  done
done
for rep in 1 2; do
timeout 300 python tools/config_bench_dist.py --chunk 256 --steps 4096 >> $O/configs_dist.jsonl 2>> $O/configs_dist.err
MXV_MIXED_SINGLE_LAUNCH=1 timeout 300 python tools/config_bench_dist.py --chunk 256 --steps 4096 >> $O/configs_dist.jsonl 2>> $O/configs_dist.err
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cd /tmp
for env in Acrobot-v1:524288 CartPole-v1:1048576 Pendulum-v1:524288 MountainCar-v0:524288 MountainCarContinuous-v0:524288; do
  e=${env%%:*}; n=${env##*:}
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_$e -o k -- \
      python $GRAFT_REPO_ROOT/tools/kbench.py --envs $e --n $n --modes fused --steps 1024 --chunk 256 > $O/pmc_$e.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$e -o k -- \
      python $GRAFT_REPO_ROOT/tools/kbench.py --envs $e --n $n --modes fused --steps 1024 --chunk 256 > $O/pmcf_$e.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$e -o k -- \
      python $GRAFT_REPO_ROOT/tools/kbench.py --envs $e --n $n --modes fused --steps 1024 --chunk 256 > $O/pmcw_$e.log 2>&1
  timeout 200 python $GRAFT_REPO_ROOT/tools/kbench.py --envs $e --n $n --modes fused,fused-final --steps 2048 --chunk 256 --tag time 2>/dev/null | grep '^{' >> $O/times.jsonl
done
cd $GRAFT_REPO_ROOT
find $O -type f -name "*.csv" ! -name "*counter_collection.csv" -delete
find $O -type f \( -name "*.db" -o -name "*.json" -o -name "*.pftrace" \) -delete
echo done > $O/finished
