#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_tab; rm -rf $out; mkdir -p $out
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o t -- python $GRAFT_REPO_ROOT/tools/tab_bench.py --ids Taxi-v3,FrozenLake-v1 --chunk 128 --reps 3 > $out/$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $GRAFT_REPO_ROOT/tools/tab_bench.py --ids Taxi-v3,FrozenLake-v1 --chunk 128 --reps 10 --tune > $out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_tab/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'tab_step_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
n=(1<<20)*128
for k,v in acc.items():
    big=[x for x in v if x>max(v)*0.5]
    m=sum(big)/len(big)
    scale=2.0 if k=='FETCH_SIZE' else 1.0   # gfx950: FETCH_SIZE counts half the bytes (tools/calib)
    print(f"{k}: {m:.0f} KiB per 128-step launch (x{scale} calibration) = {m*1024*scale/n:.2f} B per env-step over {len(big)} launches")
for f in glob.glob('gpurun_out/pmc_tab/trace/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'tab_step_kernel' in r['Name']: print('rocprof stats:', r['Name'][:40], 'calls', r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min us', float(r['MinNs'])/1e3)
PY
tail -2 gpurun_out/pmc_tab/trace.log | cut -c1-200
