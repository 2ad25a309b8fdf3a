#!/usr/bin/env bash
# Builds libmxv tuning variants into gym_amd/_lib/variants/.  Spec: name:E:E_acrobot:consec:minwaves[:extra -D flags]
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gym_amd/_lib/variants"
mkdir -p "$out"
for s in "$@"; do
  IFS=: read -r name e ea c mw extra <<<"$s"
  (
    tmp="$(mktemp -d)"
    F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMXV_ENVS_PER_LANE=$e -DMXV_ENVS_PER_LANE_ACROBOT=$ea -DMXV_CONSEC=$c -DMXV_MIN_WAVES=$mw $extra"
    /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_kernels.hip" -o "$tmp/k.o" 2>/dev/null &&
    /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_api.cpp" -o "$tmp/a.o" &&
    /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_norm.hip" -o "$tmp/n.o" 2>/dev/null &&
    /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_subnorm.hip" -o "$tmp/s.o" 2>/dev/null &&
    /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_tab.hip" -o "$tmp/t.o" 2>/dev/null &&
    /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_bj.hip" -o "$tmp/b.o" 2>/dev/null &&
    /opt/rocm/bin/hipcc $F -c "$root/gym_amd/csrc/mxv_placed.hip" -o "$tmp/p.o" 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmxv_$name.so" "$tmp/k.o" "$tmp/a.o" "$tmp/n.o" "$tmp/s.o" "$tmp/t.o" "$tmp/b.o" "$tmp/p.o" && echo "built $name"
    rm -rf "$tmp"
  ) &
done
wait
