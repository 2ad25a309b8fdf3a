#!/usr/bin/env bash
mkdir -p gpurun_out tools/_bin; export TMPDIR=/tmp
{
/opt/rocm/bin/rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -v "^$" | head -12
/opt/rocm/bin/rocm-smi --showmeminfo vram --showclocks 2>&1 | grep -v "^$" | head -14
/opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|Cacheline|L2|L3|Memory Properties|Pool Info|Size:" | head -30
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_bin/wbench tools/wbench.hip 2>/dev/null
echo "=== wbench"; timeout 300 tools/_bin/wbench | head -8
echo "=== bench"; timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -E "simd_count|mem_banks|num_xcc|array_count|cu_per" | head
} > gpurun_out/run23.log 2>&1
cat gpurun_out/run23.log
