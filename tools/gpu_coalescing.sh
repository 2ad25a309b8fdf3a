#!/usr/bin/env bash
# Coalescing and residency evidence for the headline kernel (north_star: "evidence coalescing and occupancy with rocprof"): L2 -> memory write
# requests of rollout_kernel_v3<CartPole, E = 2, OUT = 1> per 256-step launch, how many of them are full 64-byte requests, and waves per launch.
#   tools/gpu_coalescing.sh <tag>   ->  gpurun_out/coalescing_<tag>.json
TAG=${1:-r5}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_coalescing_$TAG; rm -rf $out; mkdir -p $out
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --steps 1024 --warmup 256 --headline-file $out/line.json"
timeout 400 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $out/wr -o b -- $B > $out/wr.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sq -o b -- $B > $out/sq.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 - <<'PY'
import csv, glob, json, os, collections
tag = os.environ["TAG"]
res = {}
for sub in ("wr", "sq"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_coalescing_{tag}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_kernel_v3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v[-4:]) / len(v[-4:])          # the last four dispatches: the timed region
out = {"what": "rocprofv3 --pmc (two passes, kernel-trace only) of `bench.py --no-variants --steps 1024 --warmup 256`: rollout_kernel_v3<CartPole, E = 2, OUT = 1>, "
               "mean of the last four 256-step launches over 2^20 envs", "counters": res}
if "TCC_EA0_WRREQ_sum" in res:
    w = res["TCC_EA0_WRREQ_sum"]
    out["write_requests_per_launch"] = w
    out["fraction_full_64B_requests"] = res.get("TCC_EA0_WRREQ_64B_sum", 0.0) / w
    out["bytes_if_all_64B"] = w * 64
    out["stored_bytes_per_launch_expected"] = 34 * (1 << 20) * 256 + 2 * 40 * (1 << 20)
if "SQ_WAVES" in res:
    out["waves_per_launch"] = res["SQ_WAVES"]
    if res.get("SQ_BUSY_CYCLES") and res.get("SQ_WAVE_CYCLES"):
        out["mean_waves_in_flight_per_SE_busy_cycle"] = res["SQ_WAVE_CYCLES"] / res["SQ_BUSY_CYCLES"]
json.dump(out, open(f"gpurun_out/coalescing_{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
