#!/usr/bin/env bash
# Builds libmxv variants with different envs-per-lane into gym_amd/_lib/variants/ (tuning only).
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gym_amd/_lib/variants"
mkdir -p "$out"
for spec in "${@:-E1:1:1 E2:2:1 E4:4:1 E8:8:2}"; do
  for s in $spec; do
    IFS=: read -r name e ea <<<"$s"
    (
      tmp="$(mktemp -d)"
      F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMXV_ENVS_PER_LANE=$e -DMXV_ENVS_PER_LANE_ACROBOT=$ea"
      /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_kernels.hip" -o "$tmp/k.o"
      /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_api.cpp" -o "$tmp/a.o"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmxv_$name.so" "$tmp/k.o" "$tmp/a.o"
      rm -rf "$tmp"; echo "built $out/libmxv_$name.so"
    ) &
  done
done
wait
