#!/usr/bin/env bash
# Prints VGPR/SGPR/scratch/LDS/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage).
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off ${MXV_EXTRA_FLAGS:-} -c "$here/mxv_kernels.hip" -o /tmp/mxv_ru.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?)\s*\[-Rpass", line) or re.search(r":\d+:\d+:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name") or t.startswith("Name:"):
        if cur: print(cur)
        cur=t.split(":",1)[1].strip()[:70].ljust(72)
    elif any(k in t for k in ("VGPRs:","TotalSGPRs","ScratchSize","Occupancy","LDS Size","Spill")):
        cur+=" "+t.replace(" ","")
if cur: print(cur)
'
