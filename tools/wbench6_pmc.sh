#!/usr/bin/env bash
# Counters of tools/wbench6's store kernel per dispatch (fast vs slow allocations of the same pattern): where do the write requests
# of a slow allocation wait?  --pmc with --kernel-trace only (one pass per counter set).
mkdir -p gpurun_out/r03y tools/_bin; export TMPDIR=/tmp
[ -x tools/_bin/wbench6 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_bin/wbench6 tools/wbench6.hip
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum"; do
  i=$((i+1))
  rm -rf /tmp/p$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p$i -o w -- $R/tools/_bin/wbench6 > $R/gpurun_out/r03y/run$i.log 2>&1
  cp /tmp/p$i/*counter_collection.csv $R/gpurun_out/r03y/counters$i.csv 2>/dev/null
  cp /tmp/p$i/*kernel_trace.csv $R/gpurun_out/r03y/trace$i.csv 2>/dev/null
done
cd $R
python3 - <<'PY'
import csv, collections, json
for i in (1, 2):
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/r03y/counters{i}.csv")))
        tr = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f"gpurun_out/r03y/trace{i}.csv"))}
    except Exception as e:
        print("pass", i, "unreadable", e); continue
    per = collections.defaultdict(dict)
    for r in rows:
        if "stores" not in r["Kernel_Name"]: continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    out = []
    for d, c in per.items():
        c["us"] = tr.get(d, 0.0)
        out.append(c)
    out.sort(key=lambda c: c["us"])
    print("pass", i, "dispatches", len(out))
    for c in out[:4] + out[-4:]:
        print(json.dumps({k: round(v, 1) for k, v in c.items()}))
PY
