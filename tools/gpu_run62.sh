#!/usr/bin/env bash
# the three numbers per config - (i) kernel duration + PMC bytes for every env kind, (ii) device loop, (iii) NumPy loop
mkdir -p gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_bytes; rm -rf $out; mkdir -p $out
cd /tmp
B="python $GRAFT_REPO_ROOT/tools/kbench.py --envs Pendulum-v1,MountainCarContinuous-v0,Acrobot-v1,MountainCar-v0 --n 524288 --modes fused --steps 512 --chunk 128"
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o k -- $B > $out/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections,re
acc=collections.defaultdict(list); dur=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_bytes/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(rollout_kernel(?:_v2)?)<(\d)',r['Kernel_Name'])
        if not m: continue
        k=m.group(1)+'<'+m.group(2)+'>'
        acc[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
names={'1':'Pendulum-v1','2':'Acrobot-v1','3':'MountainCar-v0','4':'MountainCarContinuous-v0'}
n=(1<<19)*128
print("# fused rollout, 2^19 envs, 128-step launches; PMC passes WRITE_SIZE / FETCH_SIZE separately (FETCH x2: gfx950 calibration, tools/calib)")
for k in sorted(dur):
    w=acc[(k,'WRITE_SIZE')]; f=acc[(k,'FETCH_SIZE')]
    wb=sum(w)/len(w)*1024/n; fb=sum(f)/len(f)*1024*2/n; d=sorted(dur[k])[len(dur[k])//2]
    print(f"{names[k[-2]]:26s} {k:24s} median launch {d:8.1f} us under PMC ({d/128:.2f} us/step)  write {wb:6.2f} B/env-step  read {fb:5.2f} B/env-step  -> {(wb+fb)*n/d/1e6:6.0f} GB/s real")
PY
timeout 600 python tools/config_bench.py 2>/dev/null | grep "^{" > gpurun_out/configs_r01j.jsonl; cut -c1-250 gpurun_out/configs_r01j.jsonl
