#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python tools/soak.py > gpurun_out/soak.log 2>&1
tail -6 gpurun_out/soak.log
