#!/usr/bin/env bash
mkdir -p gpurun_out tools/_bin; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_bin/wbench3 tools/wbench3.hip 2>&1 | grep -i "error"
for i in 1 2; do timeout 300 tools/_bin/wbench3 map; done > gpurun_out/run42b.log 2>&1
cat gpurun_out/run42b.log
