#!/usr/bin/env bash
# Memory-path counters (separate --pmc passes, kernel-trace only) for the fused CartPole rollout and for tools/wbench's
# pure-store kernels: where do write requests stall?
mkdir -p gpurun_out tools/_bin; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_mem; rm -rf $out; mkdir -p $out
[ -x $GRAFT_REPO_ROOT/tools/_bin/wbench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $GRAFT_REPO_ROOT/tools/_bin/wbench $GRAFT_REPO_ROOT/tools/wbench.hip 2>/dev/null
cd /tmp
B="python $GRAFT_REPO_ROOT/tools/kbench.py --envs CartPole-v1 --n 1048576 --modes fused --steps 1024 --chunk 256"
W="$GRAFT_REPO_ROOT/tools/_bin/wbench"
i=0
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL" \
           "TCC_WRITE TCC_WRITEBACK TCC_WRITE_SECTORS TCC_TOO_MANY_EA_WRREQS_STALL" \
           "TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_THRASHING_STALL" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_TAG_STALL TCC_IB_STALL TCC_REQ TCC_EA0_WRREQ_LEVEL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/b$i -o b -- $B > $out/b$i.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/w$i -o w -- $W > $out/w$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_mem/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'rollout_kernel' in k: k='rollout_kernel'
        elif 'traj_write' in k or 'fill' in k or 'copy' in k: k=k[:60]
        else: continue
        acc[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
names=sorted({c for _,c in acc})
kern=sorted({k for k,_ in acc})
for k in kern:
    print("==",k, "launches", len(next(iter([v for (kk,c),v in acc.items() if kk==k]))))
    for c in names:
        v=acc.get((k,c))
        if v: print(f"   {c:44s} {sum(v)/len(v):16.0f}")
PY
