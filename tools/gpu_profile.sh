#!/usr/bin/env bash
# rocprofv3 evidence for profiles/: kernel trace + stats of the bench command, then PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate runs; --pmc only ever combined with --kernel-trace) of the bench and of tools/calib.
R=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- $B --steps 1000 --warmup 100 > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o bench -- $B --steps 40 --warmup 10 --no-graph > $out/pmc_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/calib_$c -o calib -- $GRAFT_REPO_ROOT/tools/calib > $out/calib_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
find $out -type f | head -40
du -sh $out
