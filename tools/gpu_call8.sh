#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02p
mkdir -p $O
timeout 600 python tools/placement_scan.py 2>/dev/null | grep '^{' > $O/placement_scan.jsonl
V=gym_amd/_lib/variants
kb() { timeout 200 python tools/kbench.py --lib $V/libmxv_$1.so --tag $1 --envs $2 --n $3 --steps $4 --chunk 256 --modes $5 2>/dev/null | grep '^{' >> $O/lds_state_ab.jsonl; }
for rep in 1 2; do for v in v3 ldsstate; do
  kb $v CartPole-v1 1048576 4096 fused,fused-final,fusedf32
  kb $v Acrobot-v1 524288 1024 fused
  kb $v Pendulum-v1 1048576 2048 fused,fused-final
  kb $v CartPole-v1 131072 4096 fused
done; done
echo done > $O/finished
