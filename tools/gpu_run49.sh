#!/usr/bin/env bash
mkdir -p gpurun_out tools/_bin; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_bin/wbench4 tools/wbench4.hip 2>&1 | grep -i " error"
timeout 600 tools/_bin/wbench4 > gpurun_out/run49.log 2>&1
python3 - <<'PY'
import re
for line in open("gpurun_out/run49.log"):
    if line.startswith("#"): print(line.strip()); continue
    v=[(float(a),float(b)) for a,b in re.findall(r"(\d+):([\d.]+)", line)]
    if not v: continue
    ts=[b for _,b in v]
    print(f"  n={len(v)} min={min(ts):.2f} max={max(ts):.2f} mean={sum(ts)/len(ts):.2f}")
    slow=[int(a) for a,b in v if b>min(ts)*1.07]
    print("  slow (>7% over min) at:", slow[:80])
PY
