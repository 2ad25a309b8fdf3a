#!/usr/bin/env bash
# SQ_INSTS_VALU / SQ_WAVES of the Blackjack trajectory kernel (bj_kernel), both dtype sets: VALU instructions per table-step, measured.
#   tools/gpu_valu_bj.sh <tag>   ->  gpurun_out/valu_bj_<tag>.json
TAG=${1:-r5}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu_bj_$TAG; rm -rf $out; mkdir -p $out
cat > /tmp/bj_run.py <<PY
import sys, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from gym_amd.toy_text import BlackjackRollout
compact = sys.argv[1] == "compact"
r = BlackjackRollout(1 << 20, seed=0, action_seed=1, compact=compact)
r.reset(seed=0)
out = r.trajectory_buffers(128, layout="separate")
for _ in range(6):
    r.rollout_per_step(128, out=out)
r.synchronize()
PY
cd /tmp
for v in ref compact; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $out/$v -o b -- python /tmp/bj_run.py $v > $out/$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 - <<'PY'
import csv, glob, json, os, collections
tag = os.environ["TAG"]
res = {}
for v in ("ref", "compact"):
    acc, kname = collections.defaultdict(list), None
    for f in glob.glob(f"gpurun_out/pmc_valu_bj_{tag}/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bj_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kname = r["Kernel_Name"]
    if not acc:
        continue
    m = {k: sum(x) / len(x) for k, x in acc.items()}
    res[v] = {"kernel": kname.split("(")[0][-48:], "launches_averaged": len(acc["SQ_INSTS_VALU"]), "steps_per_launch": 128, "waves": m["SQ_WAVES"],
              "valu_per_table_step": m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / 128, "salu_per_wave_step": m["SQ_INSTS_SALU"] / m["SQ_WAVES"] / 128}
json.dump({"what": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace of BlackjackRollout(2^20 tables).rollout_per_step(128): "
                   "VALU instructions a wave issues per step of each of its 64 tables (one table per lane)", "tag": tag, "variants": res},
          open(f"gpurun_out/valu_bj_{tag}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
