#!/usr/bin/env bash
# What does a step(actions) launch really move?  A/B of the loop (default vs MXV_ELAPSED32=1, alternating in ONE box), then
# separate --pmc FETCH_SIZE / WRITE_SIZE passes (kernel-trace only) over 200 bare launches in both dtype sets, calibrated on tools/calib.
#   tools/gpu_step_traffic.sh <tag>          -> gpurun_out/<tag>/{ab.jsonl,traffic.json}        (profiles/r6/r6b_*)
#   tools/gpu_step_traffic.sh <tag> shape    the four forms of the step instead (ordinary, compact, observation carries the state, both);
#                                            the A/B is tools/ab_step_shape.sh's                        (profiles/r6/r6i_*)
R=${1:-r6b}; MODE=${2:-elapsed}; O=$GRAFT_REPO_ROOT/gpurun_out/$R; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
[ -x tools/calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/calib tools/calib.hip
if [ $MODE = shape ]; then
  VARIANTS="ref compact obs compactobs"
  bash tools/ab_step_shape.sh > $O/ab.jsonl 2> $O/ab.err
else
  VARIANTS="ref ref32 compact compact32"
  for rep in 1 2 3; do
    for c in "" "--compact"; do
      python tools/step_loop.py $c --tag default >> $O/ab.jsonl 2>> $O/ab.err
      MXV_ELAPSED32=1 python tools/step_loop.py $c --tag elapsed32 >> $O/ab.jsonl 2>> $O/ab.err
    done
  done
fi
unset MXV_LIB_PATH
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for v in $VARIANTS; do
    F=""; E32=""
    case $v in compact*) F="--compact";; esac
    case $v in *obs) F="$F --obs-state";; esac
    case $v in *32) E32=1;; esac
    env ${E32:+MXV_ELAPSED32=1} timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${v}_$c -o s -- python $GRAFT_REPO_ROOT/tools/step_loop.py --plain --steps 200 $F > $O/pmc_${v}_$c.log 2>&1
  done
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -o calib -- $GRAFT_REPO_ROOT/tools/calib > $O/calib_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - $O $VARIANTS <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
def means(pattern, key):
    acc = []
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc.append(float(r["Counter_Value"]))
    return (sum(acc) / len(acc), len(acc)) if acc else (None, 0)
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    m, _ = means(f"{O}/calib_{c}/**/*counter_collection.csv", "copy8")
    if m is None: m, _ = means(f"{O}/calib_{c}/*counter_collection.csv", "copy8")
    cal[c] = None if m is None else m * 1024 / (64 << 20)
out = {"calibration_ratio_on_64MiB_copies": cal, "envs": 1 << 20, "variants": {}}
for v in sys.argv[2:]:
    row = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        m, n = means(f"{O}/pmc_{v}_{c}/**/*counter_collection.csv", "step_kernel")
        if m is None: m, n = means(f"{O}/pmc_{v}_{c}/*counter_collection.csv", "step_kernel")
        row[c + "_KiB_per_launch"] = m; row[c + "_launches"] = n
        row[c + "_bytes_per_env_step"] = None if (m is None or not cal[c]) else m * 1024 / cal[c] / (1 << 20)
    if row["FETCH_SIZE_bytes_per_env_step"] and row["WRITE_SIZE_bytes_per_env_step"]:
        row["bytes_per_env_step"] = row["FETCH_SIZE_bytes_per_env_step"] + row["WRITE_SIZE_bytes_per_env_step"]
    out["variants"][v] = row
json.dump(out, open(f"{O}/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -type f -name "*.csv" ! -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete 2>/dev/null
cat $O/ab.jsonl
