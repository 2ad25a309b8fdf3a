#!/usr/bin/env bash
# Box characterisation: VALU issue rates (ratebench), write ceilings (wbench) and the bench line on the SAME box.
mkdir -p gpurun_out tools/_bin; export TMPDIR=/tmp
{
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>&1 | grep -v "^$" | head -20
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ratebench tools/ratebench.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_bin/wbench tools/wbench.hip 2>/dev/null
echo "=== ratebench"; timeout 300 tools/_bin/ratebench
echo "=== wbench"; timeout 300 tools/_bin/wbench
echo "=== bench"; timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>&1 | grep -v "^$" | head -20
} > gpurun_out/run12.log 2>&1
tail -c 3000 gpurun_out/run12.log
