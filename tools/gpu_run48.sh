#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 2048000 --warmup 2048 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('12-second run:', round(j['value']/1e9,1), 'Ge/s', round(j['ms_per_step']*1e3,3), 'us/step, frac', round(j['roofline']['frac'],3), 'total env-steps', j['steps']*2**20)"
