#!/usr/bin/env bash
# Builds gym_amd/_lib/libmxv.so for gfx950 (cross-compiles without a GPU).
#   -ffp-contract=off : the reference rounds every operation; never fuse a*b+c into an FMA.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../_lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result ${MXV_EXTRA_FLAGS:-})
"$HIPCC" "${FLAGS[@]}" -c "$here/mxv_kernels.hip" -o "$out/mxv_kernels.o" &
"$HIPCC" "${FLAGS[@]}" -c "$here/mxv_api.cpp" -o "$out/mxv_api.o" &
"$HIPCC" "${FLAGS[@]}" -c "$here/mxv_norm.hip" -o "$out/mxv_norm.o" &
"$HIPCC" "${FLAGS[@]}" -c "$here/mxv_tab.hip" -o "$out/mxv_tab.o" &
"$HIPCC" "${FLAGS[@]}" -c "$here/mxv_bj.hip" -o "$out/mxv_bj.o" &
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out/libmxv.so" "$out/mxv_kernels.o" "$out/mxv_api.o" "$out/mxv_norm.o" "$out/mxv_tab.o" "$out/mxv_bj.o"
echo "built $out/libmxv.so"
