#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/run61_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/run61_tests.log
tail -3 gpurun_out/run61_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py 2>/dev/null | tail -1 | cut -c1-330
