#!/usr/bin/env bash
# rocprofv3 kernel stats of every per-config kernel (tools/config_bench.py), the tabular/Blackjack engines and the normalisers
mkdir -p gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg; rm -rf $out; mkdir -p $out
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg -o c -- python $GRAFT_REPO_ROOT/tools/config_bench.py > $out/cfg.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/norm -o n -- python $GRAFT_REPO_ROOT/tools/norm_bench.py > $out/norm.log 2>&1
cd $GRAFT_REPO_ROOT
find $out -type f ! -name "*kernel_stats.csv" ! -name "*.log" -delete
grep -h "^{" $out/cfg.log | cut -c1-260
grep -h "^{" $out/norm.log | cut -c1-260 | head
head -12 $out/cfg/*kernel_stats.csv | cut -c1-200
