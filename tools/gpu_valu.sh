#!/usr/bin/env bash
# SQ_INSTS_VALU / SQ_WAVES of the fused trajectory kernel of every env kind (one --pmc pass per kind, kernel-trace only)
#   tools/gpu_valu.sh <tag> [lib variant]     ->  gpurun_out/valu_<tag>.json   (copy to profiles/: bench.py's roofline_valu reads it)
TAG=${1:-r4a}; LIBARG=""; [ -n "$2" ] && LIBARG="--lib $GRAFT_REPO_ROOT/gym_amd/_lib/variants/libmxv_$2.so"
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu_$TAG; rm -rf $out; mkdir -p $out
cd /tmp
for spec in CartPole-v1:1048576 Pendulum-v1:524288 Acrobot-v1:524288 MountainCar-v0:524288 MountainCarContinuous-v0:524288; do
  e=${spec%%:*}; n=${spec##*:}
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/$e -o b -- \
    python $GRAFT_REPO_ROOT/tools/kbench.py $LIBARG --envs $e --n $n --modes fused --steps 512 --chunk 256 --layout separate > $out/$e.log 2>&1
done
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 - <<'PY'
import csv, glob, json, os, collections
tag = os.environ["TAG"]
EPL = {"CartPole-v1": 2, "Pendulum-v1": 1, "Acrobot-v1": 1, "MountainCar-v0": 2, "MountainCarContinuous-v0": 2}
kinds = {}
for d in sorted(glob.glob(f"gpurun_out/pmc_valu_{tag}/*/")):
    env = os.path.basename(d.rstrip("/"))
    acc = collections.defaultdict(list)
    kname = None
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_kernel_v3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kname = r["Kernel_Name"]
    if not acc:
        continue
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    steps = 256
    kinds[env] = {"kernel": kname.split("(")[0][-60:], "launches_averaged": len(acc["SQ_INSTS_VALU"]), "steps_per_launch": steps,
                  "envs_per_lane": EPL[env], "waves": m["SQ_WAVES"], "valu_per_wave_step": m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / steps,
                  "salu_per_wave_step": m["SQ_INSTS_SALU"] / m["SQ_WAVES"] / steps, "lds_per_wave_step": m["SQ_INSTS_LDS"] / m["SQ_WAVES"] / steps}
json.dump({"what": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace of tools/kbench.py --modes fused --chunk 256 "
                   "(the trajectory launch bench.py times), averaged over the launches of the run; per wave and step (a wave steps "
                   "64 x envs_per_lane envs)", "tag": tag, "kinds": kinds}, open(f"gpurun_out/valu_{tag}.json", "w"), indent=1)
print(json.dumps(kinds, indent=1))
PY
