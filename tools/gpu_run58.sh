#!/usr/bin/env bash
# Instruction mix and issue statistics of the fused rollout per env kind (trajectory mode and cache-resident "final" mode)
mkdir -p gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_kinds; rm -rf $out; mkdir -p $out
cd /tmp
B="python $GRAFT_REPO_ROOT/tools/kbench.py --envs Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0,CartPole-v1 --n 1048576 --modes fused,fused-final --steps 512 --chunk 128"
$B 2>&1 | grep "^{" | cut -c1-220
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o k -- $B > $out/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections,re
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_kinds/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        m=re.search(r'(rollout_kernel(?:_v2)?)<(\d)',k)
        if not m: continue
        acc[(m.group(1)+'<'+m.group(2)+'>',r['Counter_Name'])].append(float(r['Counter_Value']))
kern=sorted({k for k,_ in acc}); names=sorted({c for _,c in acc})
n=(1<<20)*128
print('per env-step (128-step launches, 2^20 envs; means over all launches of both modes):')
print('counter'.ljust(28)+''.join(k.rjust(24) for k in kern))
for c in names:
    print(c.ljust(28)+''.join(f"{(sum(acc[(k,c)])/len(acc[(k,c)])*64/n if (k,c) in acc else float('nan')):24.2f}" for k in kern))
PY
