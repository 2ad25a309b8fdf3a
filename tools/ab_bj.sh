#!/usr/bin/env bash
# Blackjack rollout at 2^20 tables for the default library and every variant in gym_amd/_lib/variants/, alternating, in ONE box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for f in "" gym_amd/_lib/variants/libmxv_*.so; do
    if [ -z "$f" ]; then unset MXV_LIB_PATH; v=default; else [ -e "$f" ] || continue; export MXV_LIB_PATH=$GRAFT_REPO_ROOT/$f; v=${f##*libmxv_}; v=${v%.so}; fi
    python - <<PY
import torch, bench
print("variant=$v", "blackjack", round(bench.measure_blackjack(torch, 1<<20, 128)["us_per_step"],3))
PY
  done
done
