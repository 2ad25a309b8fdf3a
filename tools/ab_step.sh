#!/usr/bin/env bash
# step(actions) at 2^20 envs for the default library and every variant in gym_amd/_lib/variants/, alternating, in ONE box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for f in "" gym_amd/_lib/variants/libmxv_*.so; do
    if [ -z "$f" ]; then unset MXV_LIB_PATH; v=default; else [ -e "$f" ] || continue; export MXV_LIB_PATH=$GRAFT_REPO_ROOT/$f; v=${f##*libmxv_}; v=${v%.so}; fi
    python - <<PY
import torch
from benchmarks.loops import measure_step_loop, measure_step_kernel
print("variant=$v", "step_loop", round(measure_step_loop(torch, 1<<20)["us_per_step"],2), "kernel", round(measure_step_kernel(torch, 1<<20)["us_per_launch_median"],2))
PY
  done
done
