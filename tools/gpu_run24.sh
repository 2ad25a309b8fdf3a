#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 120 rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/avail.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -c . gpurun_out/avail.txt
grep -o "Name:[^,]*\(SQ_\|TCC_\|TCP_\|TA_\|TD_\|GRBM_\)[A-Z0-9_]*" gpurun_out/avail.txt | sed 's/.*Name:\s*//' | sort -u | tr '\n' ' ' | head -c 9000
