#!/usr/bin/env bash
# Instruction-mix counters of the fused rollout kernel (separate --pmc passes, kernel-trace only).
mkdir -p gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_inst; rm -rf $out; mkdir -p $out
cd /tmp
ENVID=${1:-CartPole-v1}; NENV=${2:-1048576}
LIBARG=""; [ -n "$3" ] && LIBARG="--lib $GRAFT_REPO_ROOT/gym_amd/_lib/variants/libmxv_$3.so"
B="python $GRAFT_REPO_ROOT/tools/kbench.py $LIBARG --envs $ENVID --n $NENV --modes fused --steps 200 --chunk 100"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o b -- $B > $out/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
NENV=$NENV python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_inst/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'rollout_kernel' in r['Kernel_Name'] or 'step_kernel<' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    m=sum(v)/len(v)
    import os
    nenv=int(os.environ.get('NENV','1048576'))
    print(f"{k:32s} {m:16.0f} per launch  {m/(100*nenv)*64:10.2f} per wave-env-step (x64/env-step)")
PY
