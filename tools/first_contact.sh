#!/usr/bin/env bash
# First contact with a multi-GPU node, read in one glance.  Runs, over RCCL (backend nccl), on the GPUs the node has:
#     python bench.py --gpus 1           the N = 1 line (the efficiency denominator)
#     python bench.py --gpus 2 | 4 | 8   bench.py starts its own ranks (as the driver's torch.distributed.run does)
#     pytest tests/test_gpu_comm.py      the C ABI's own RCCL communicator (mxv_comm_*)
# and prints ONE table: ranks_seen, gathers in the timed region, the gather alone (measured vs the xGMI link model), every rank's kernel
# time and placement walk, env-steps/s and the efficiency against N x the one-GPU line.  Nothing here is measured by gpurun's one GPU:
# it is the script to run the day an 8-GPU node is there.        usage: tools/first_contact.sh [steps] [warmup]
cd "$(dirname "$0")/.." || exit 1
STEPS=${1:-20}; WARMUP=${2:-5}
OUT=${FIRST_CONTACT_OUT:-gpurun_out/first_contact}; mkdir -p "$OUT"
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "devices visible: $NGPU"
for n in 1 2 4 8; do
  [ "$n" -le "$NGPU" ] || { echo "skipping --gpus $n (only $NGPU device(s))"; continue; }
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python bench.py --gpus $n --steps $STEPS --warmup $WARMUP --no-variants --no-cpu-baseline \
      > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err" || echo "bench --gpus $n failed (rc $?): tail of $OUT/bench_$n.err:" "$(tail -3 "$OUT/bench_$n.err")"
done
if [ "$NGPU" -ge 2 ]; then
  timeout 900 python -m pytest tests/test_gpu_comm.py -m gpu -q > "$OUT/pytest_comm.log" 2>&1; echo "tests/test_gpu_comm.py: $(tail -1 "$OUT/pytest_comm.log")"
fi
python - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
rows, base = [], None
for n in (1, 2, 4, 8):
    p = os.path.join(out, f"bench_{n}.json")
    try:
        line = json.loads([l for l in open(p) if l.startswith("{")][-1])
    except Exception:
        continue
    c = line["config"]
    if n == 1:
        base = line["value"]
    g = c.get("gather_us") or {}
    fields = c.get("per_rank_fields", [])
    col = {k: i for i, k in enumerate(fields)}
    kern = [r[col["kernel_us_per_step"]] for r in c.get("per_rank", [])]
    walk = [r[col["placement_seconds"]] for r in c.get("per_rank", [])] if "placement_seconds" in col else []
    rows.append((n, c.get("ranks_seen"), c.get("gathers_in_timed_region"), g.get("measured_blocking"), g.get("predicted"),
                 (min(kern), max(kern)) if kern else None, max([w for w in walk if w is not None], default=None), line["value"],
                 None if not base else line["value"] / (base * (1 if line.get("scaling") == "strong" else n))))
print(f"{'gpus':>4} {'ranks':>5} {'gathers':>7} {'gather us (measured | model)':>32} {'kernel us/step (min..max)':>27} {'walk s':>7} {'env-steps/s':>12} {'vs N=1':>7}")
for n, ranks, gathers, gm, gp, kern, walk, value, eff in rows:
    gtxt = "-" if gm is None else f"{gm:9.1f} | {gp[0]:.0f}..{gp[1]:.0f}" if gp else f"{gm:9.1f}"
    ktxt = "-" if kern is None else f"{kern[0]:.3f}..{kern[1]:.3f}"
    print(f"{n:>4} {str(ranks):>5} {str(gathers):>7} {gtxt:>32} {ktxt:>27} {('-' if walk is None else f'{walk:.2f}'):>7} {value:12.4g} {('-' if eff is None else f'{eff:.2f}x'):>7}")
print("(strong scaling: 'vs N=1' is the speed-up over the one-GPU line, target >= 6x at 8 GPUs; the gather model is 7 links x 80-150 GB/s)")
PY
