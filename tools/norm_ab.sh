#!/usr/bin/env bash
# tools/norm_ab.py for the default library and for every variant in gym_amd/_lib/variants/ (tools/build_variants.sh), twice, in ONE box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4n
for rep in 1 2; do
  unset MXV_LIB_PATH; echo "variant=default $(python tools/norm_ab.py 2>/dev/null | tail -1)"
  for f in gym_amd/_lib/variants/libmxv_*.so; do
    [ -e "$f" ] || continue
    export MXV_LIB_PATH=$GRAFT_REPO_ROOT/$f; v=${f##*libmxv_}; echo "variant=${v%.so} $(python tools/norm_ab.py 2>/dev/null | tail -1)"
  done
done
