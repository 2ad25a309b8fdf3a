#!/usr/bin/env bash
# What launch shape should one step(actions) of CartPole at 2^20 envs have?  Alternating runs inside ONE box of
#   default  = the built library: one env per lane from 2^19 envs (two rounds of 8 waves per SIMD), two per lane below
#   e2       = two envs per lane at every size (one round of 8 waves per SIMD: what rounds 3-5 shipped)
# each in the four forms of the step: ordinary, compact outputs, the observation carries the state, both.
#   st0      = the launch's stores without the nontemporal hint (-DMXV_STEP_NT_STORES=0)
# Build the comparison libraries first (here, on the CPU; every library under gym_amd/_lib/variants/ is run):
#   tools/build_variants.sh 'e2:2:1:0:1:-DMXV_STEP_E1_FROM=((int64_t)1<<60)' st0:2:1:0:1:-DMXV_STEP_NT_STORES=0
# then:  gpurun --timeout 900 -- 'bash tools/ab_step_shape.sh > gpurun_out/ab_step_shape.txt'
# -> profiles/r6/r6i_step_launch_shape.md
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for rep in 1 2 3; do
for v in default $(ls gym_amd/_lib/variants 2>/dev/null | sed -n 's/^libmxv_\(.*\)\.so$/\1/p'); do
  if [ "$v" = default ]; then unset MXV_LIB_PATH; else export MXV_LIB_PATH=$PWD/gym_amd/_lib/variants/libmxv_$v.so; fi
  for flags in "" "--compact" "--obs-state" "--compact --obs-state"; do
    echo "variant=$v flags='$flags' $(python tools/step_loop.py --envs 1048576 --steps 400 $flags | tail -1)"
  done
done; done
