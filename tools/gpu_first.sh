#!/usr/bin/env bash
# First contact with the MI355X: tests, smoke, bench, variant sweep, rocprof kernel trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8; nproc
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "=== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30
echo "=== bench"; timeout 600 python bench.py --steps 1000 --warmup 100 2>&1 | tail -5
echo "=== bench eager"; timeout 600 python bench.py --steps 1000 --warmup 100 --no-graph --no-cpu-baseline 2>&1 | tail -3
echo "=== kbench default lib all envs"; timeout 600 python tools/kbench.py --envs CartPole-v1,Pendulum-v1,MountainCar-v0,MountainCarContinuous-v0 --n 1048576 --modes graph,given,f32 2>&1 | tail -20
timeout 300 python tools/kbench.py --envs Acrobot-v1 --n 524288 --modes graph,given 2>&1 | tail -5
for v in E1 E2 E4 E8; do echo "=== variant $v"; timeout 300 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs CartPole-v1,Pendulum-v1 --modes graph 2>&1 | tail -3; timeout 200 python tools/kbench.py --lib gym_amd/_lib/variants/libmxv_$v.so --tag $v --envs Acrobot-v1 --n 524288 --modes graph 2>&1 | tail -2; done
echo "=== rocprof kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>&1 | tail -5
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_r01 -name "*stats*" | head; 
} > gpurun_out/first.log 2>&1
tail -c 6000 gpurun_out/first.log
