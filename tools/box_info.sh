#!/usr/bin/env bash
# What kind of box is this?  rocm-smi state next to the placement probe (fast / slow kind).
O=${1:-gpurun_out/box_info.txt}
{
echo "== date $(date -u +%FT%TZ) host $(hostname)"
rocm-smi --showclocks --showperflevel --showpower --showmaxpower --showmemuse --showmeminfo vram --showvoltage --showtemp 2>&1 | grep -v "^$" | head -60
rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -v "^$" | head -12
rocm-smi --showrasinfo all 2>&1 | grep -iv "^$" | head -30
rocm-smi --showretiredpages 2>&1 | grep -v "^$" | head -12
rocm-smi --showpids 2>&1 | grep -v "^$" | head -8
cat /sys/class/drm/card*/device/mem_info_vram_used 2>/dev/null | head -2
cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition 2>/dev/null
} > $O 2>&1
timeout 300 python tools/placement_scan2.py 2>/dev/null | grep '^{' | head -1 >> $O
