#!/usr/bin/env bash
# rocprofv3 evidence for profiles/: kernel trace + stats of the bench command, then PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate runs; --pmc only ever combined with --kernel-trace) of the bench and of tools/calib.
#   tools/gpu_profile.sh <tag> <mode> <chunk>      e.g.  r01f fused 100   |   r01 eager 100
R=${1:-r01f}; MODE=${2:-fused}; CHUNK=${3:-100}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
[ -x $GRAFT_REPO_ROOT/tools/calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $GRAFT_REPO_ROOT/tools/calib $GRAFT_REPO_ROOT/tools/calib.hip
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --mode $MODE --chunk $CHUNK"
if [ "$MODE" = fused ]; then TS=$((CHUNK*8)); TW=$CHUNK; PS=$((CHUNK*4)); PW=$CHUNK; else TS=1000; TW=100; PS=40; PW=10; fi
echo "$MODE $CHUNK trace:$TS/$TW pmc:$PS/$PW" > $out/meta.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- $B --steps $TS --warmup $TW > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o bench -- $B --steps $PS --warmup $PW > $out/pmc_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/calib_$c -o calib -- $GRAFT_REPO_ROOT/tools/calib > $out/calib_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
# keep only what tools/summarize_profile.py reads (the raw traces are large)
find $out -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" ! -name meta.txt ! -path "*/trace/*kernel_trace.csv" -delete
tail -2 $out/trace.log
du -sh $out
