#!/usr/bin/env bash
# Blackjack rollout at 2^20 tables for the default library and every variant in gym_amd/_lib/variants/, alternating, in ONE box;
# both dtype sets, trajectory tensors sorted by HBM class (BlackjackRollout.trajectory_buffers) and ordinary ones.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for f in "" gym_amd/_lib/variants/libmxv_*.so; do
    if [ -z "$f" ]; then unset MXV_LIB_PATH; v=default; else [ -e "$f" ] || continue; export MXV_LIB_PATH=$GRAFT_REPO_ROOT/$f; v=${f##*libmxv_}; v=${v%.so}; fi
    python - <<PY 2>&1 | grep -v amdgpu.ids
import os, torch
from benchmarks.toy_text import measure_blackjack
row = ["variant=$v"]
for placement in ("on", "off"):
    os.environ["MXV_PLACEMENT"] = placement
    for compact in (False, True):
        r = measure_blackjack(torch, 1 << 20, 128, compact=compact)
        row.append(f"{'compact' if compact else 'ref'}/{'sorted' if placement == 'on' else 'plain'} {r['us_per_step']:.3f}")
print("  ".join(row))
PY
  done
done
