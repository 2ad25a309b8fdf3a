#!/usr/bin/env bash
mkdir -p gpurun_out tools/_bin; export TMPDIR=/tmp
{
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/_bin/divcheck tools/divcheck.hip 2>/dev/null
echo "=== divcheck"; timeout 300 tools/_bin/divcheck
echo "=== pytest gpu (all)"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "=== kbench acrobot"; timeout 300 python tools/kbench.py --envs Acrobot-v1 --n 524288 --modes fused,graph --steps 400 --chunk 100 2>&1 | tail -2
} > gpurun_out/run15.log 2>&1
bash tools/gpu_pmc.sh Acrobot-v1 524288 > gpurun_out/pmc_acrobot.txt 2>&1
tail -c 3000 gpurun_out/run15.log; cat gpurun_out/pmc_acrobot.txt | tail -40
