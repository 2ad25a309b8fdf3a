#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
{
echo "uid: $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)  vbios: $(cat /sys/class/drm/card*/device/vbios_version 2>/dev/null | head -1)"
echo "mem_part: $(cat /sys/class/drm/card*/device/current_memory_partition 2>/dev/null | head -1)  comp_part: $(cat /sys/class/drm/card*/device/current_compute_partition 2>/dev/null | head -1)"
cat /sys/class/drm/card*/device/pp_dpm_mclk 2>/dev/null | head -4 | tr '\n' ' '; echo
cat /sys/class/drm/card*/device/pp_dpm_fclk 2>/dev/null | head -4 | tr '\n' ' '; echo
timeout 300 python bench.py --no-cpu-baseline --no-variants 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); p=j['config']['placement']; print('bench', round(j['value']/1e9,1), 'Ge/s', round(j['ms_per_step']*1e3,3), 'us; candidates', min(p['us_per_step']), max(p['us_per_step']), 'mixes', min(p['mixes_us_per_step']), max(p['mixes_us_per_step']))"
} 2>&1 | tee -a gpurun_out/boxes.log
