#!/usr/bin/env python3
"""VGPR / SGPR / LDS / occupancy table of the kernels in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py gym_amd/csrc/mxv_kernels.hip [filter] [-- extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "--" else ""
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|TotalSGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
for r in rows:
    if flt in r["name"]:
        name = r["name"].replace("mxv::(anonymous namespace)::", "").replace("(mxv::StepArgs)", "").replace("void ", "")
        print(f'{name:60s} VGPR {str(r.get("VGPRs")):>4} SGPR {str(r.get("SGPRs")):>4} scratch {str(r.get("ScratchSize")):>4} LDS {str(r.get("LDS")):>6} occ {r.get("Occupancy")}')
