#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
