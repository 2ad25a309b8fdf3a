#!/usr/bin/env bash
# Are the placement speed modes (5.9 / 6.7 / 7.1 us per step for the same code) address-translation effects?
# Per-dispatch correlation of the fused rollout's duration with the TCP UTCL1 (L1 TLB) counters across the tuner's
# candidate sets, which differ only in where the trajectory tensors sit.
mkdir -p gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_tlb; rm -rf $out; mkdir -p $out
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --steps 1024 --warmup 256"
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_THRASHING_STALL_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/s$i -o b -- $B > $out/s$i.log 2>&1
  tail -1 $out/s$i.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_tlb/*/*counter_collection.csv')):
    rows=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'rollout_kernel' not in r['Kernel_Name']: continue
        d=rows[int(r['Dispatch_Id'])]
        d['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        d[r['Counter_Name']]=float(r['Counter_Value'])
    names=sorted({k for d in rows.values() for k in d if k!='dur'})
    print('==',f, len(rows),'rollout launches')
    print('   dur_us  '+'  '.join(n.replace('TCP_UTCL1_','')[:28].rjust(28) for n in names))
    L=sorted(rows.values(), key=lambda d:d['dur'])
    step=max(1,len(L)//24)
    for d in L[::step]:
        print(f"{d['dur']:9.1f}  "+'  '.join(f"{d.get(n,0):28.0f}" for n in names))
    import math
    for n in names:
        xs=[d['dur'] for d in L]; ys=[d.get(n,0) for d in L]
        mx=sum(xs)/len(xs); my=sum(ys)/len(ys)
        sx=math.sqrt(sum((x-mx)**2 for x in xs)); sy=math.sqrt(sum((y-my)**2 for y in ys))
        c=sum((x-mx)*(y-my) for x,y in zip(xs,ys))/(sx*sy) if sx*sy else float('nan')
        print(f"   corr(dur,{n}) = {c:.3f}")
PY
