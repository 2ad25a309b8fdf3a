#!/usr/bin/env bash
# Blackjack-only library variants: mxv_bj.hip recompiled with extra -D flags, linked with the other objects of the built library.
#   tools/build_bj_variants.sh name:"-DMXV_BJ_TILEMAP=0" ...   ->  gym_amd/_lib/variants/libmxv_<name>.so   (tools/ab_bj.sh alternates them)
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
lib="$root/gym_amd/_lib"
out="$lib/variants"
mkdir -p "$out"
for s in "$@"; do
  name="${s%%:*}"; extra="${s#*:}"
  (
    tmp="$(mktemp -d)"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra -c "$root/gym_amd/csrc/mxv_bj.hip" -o "$tmp/b.o" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmxv_$name.so" "$lib/mxv_kernels.o" "$lib/mxv_api.o" "$lib/mxv_norm.o" "$lib/mxv_tab.o" "$tmp/b.o" "$lib/mxv_placed.o" && echo "built $name"
    rm -rf "$tmp"
  ) &
done
wait
