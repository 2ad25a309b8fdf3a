#!/usr/bin/env python3
"""Condense gpurun_out/prof_<round>/ (tools/gpu_profile.sh) into the committed evidence under profiles/:

  profiles/<round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command (verbatim)
  profiles/<round>_pmc.csv            per-kernel mean FETCH_SIZE / WRITE_SIZE (raw counter units = KiB)
  profiles/traffic_<round>.json       HBM bytes per step-kernel launch after the gfx950 calibration
  profiles/<round>_summary.md         human-readable digest

Calibration (MI355X_MICROARCH.md §HBM): the counters are calibrated on tools/calib's copy kernels, which move a
known 64 MiB each way with the engine's own access widths (8 B/lane fp64 state, 16 B/lane obs, 4 B/lane elapsed,
1 B/lane flags).  corrected_read = FETCH_SIZE*1024 / fetch_ratio(copy8), corrected_write = WRITE_SIZE*1024 /
write_ratio(copy8), where ratio = counter*1024 / 67108864 on the copy kernel.
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
try:  # "<mode> <chunk> trace:<steps>/<warmup> pmc:<steps>/<warmup>" written by tools/gpu_profile.sh
    META = open(os.path.join(src, "meta.txt")).read().split()
except OSError:
    META = ["eager", "100", "trace:1000/100", "pmc:40/10"]
MODE, CHUNK = META[0], int(META[1])
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
CALIB_BYTES = 64 << 20


def counter_means(path):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def short(name):
    for key in ("step_kernel", "rollout_kernel", "reset_kernel", "sample_kernel", "copy8", "copy16", "copy4", "copy1"):
        if key in name:
            i = name.find(key)
            j = name.find("(", i)
            return name[i:j if j > 0 else None]
    return name[:60]


shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, f"{R}_kernel_stats.csv"))
stats = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    pmc[c] = {"bench": counter_means(os.path.join(src, f"pmc_{c}", "bench_counter_collection.csv")),
              "calib": counter_means(os.path.join(src, f"calib_{c}", "calib_counter_collection.csv"))}

rows = []
for where in ("bench", "calib"):
    names = sorted(set(pmc["FETCH_SIZE"][where]) | set(pmc["WRITE_SIZE"][where]))
    for k in names:
        f = pmc["FETCH_SIZE"][where].get(k, (float("nan"), 0))
        w = pmc["WRITE_SIZE"][where].get(k, (float("nan"), 0))
        rows.append({"run": where, "kernel": short(k), "launches": f[1], "FETCH_SIZE_KiB": round(f[0], 2),
                     "WRITE_SIZE_KiB": round(w[0], 2)})
with open(os.path.join(dst, f"{R}_pmc.csv"), "w", newline="") as f:
    wr = csv.DictWriter(f, fieldnames=list(rows[0]))
    wr.writeheader()
    wr.writerows(rows)


def find(where, counter, key):
    for k, v in pmc[counter][where].items():
        if key in k:
            return v[0]
    return None


ratios = {}
for key in ("copy8", "copy16", "copy4", "copy1"):
    fr = find("calib", "FETCH_SIZE", key)
    wrt = find("calib", "WRITE_SIZE", key)
    ratios[key] = {"fetch_ratio": fr * 1024 / CALIB_BYTES, "write_ratio": wrt * 1024 / CALIB_BYTES}
step_f = find("bench", "FETCH_SIZE", "step_kernel") or find("bench", "FETCH_SIZE", "rollout_kernel")
step_w = find("bench", "WRITE_SIZE", "step_kernel") or find("bench", "WRITE_SIZE", "rollout_kernel")
raw = (step_f + step_w) * 1024
corr_r = step_f * 1024 / ratios["copy8"]["fetch_ratio"]
corr_w = step_w * 1024 / ratios["copy8"]["write_ratio"]
step_stat = next(r for r in stats if "step_kernel" in r["Name"] or "rollout_kernel" in r["Name"])
# The bench command also runs the placement tuning, the spin-up and the warm-up on the same kernel: the figure comparable with
# the bench line's avg_launch_us is the mean over the dispatches of the TIMED region = the last steps/chunk launches.
timed_avg = timed_n = None
bench_line = None
try:
    ts_steps = int(META[2].split(":")[1].split("/")[0])
    n_timed = ts_steps // CHUNK if MODE == "fused" else ts_steps
    tr_path = next(os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(src, "trace")) for f in fs if f.endswith("kernel_trace.csv"))
    disp = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr_path))
                   if "step_kernel" in r["Kernel_Name"] or "rollout_kernel" in r["Kernel_Name"]))
    last = disp[-n_timed:]
    timed_avg, timed_n = sum(e - s for s, e in last) / len(last), len(last)
    for line in open(os.path.join(src, "trace.log")):
        if line.startswith("{") and '"roofline"' in line:
            bench_line = json.loads(line)
except Exception as exc:  # older profile directories hold no per-dispatch trace
    print("no timed-region figure:", exc)

traffic = {
    "round": R,
    "mode": MODE,
    "chunk": CHUNK,
    "env_steps_per_launch": (CHUNK if MODE == "fused" else 1) << 20,
    "kernel": short(step_stat["Name"]),
    "avg_launch_ns_rocprof": float(step_stat["AverageNs"]),
    "calls": int(step_stat["Calls"]),
    "avg_launch_ns_rocprof_timed_region": timed_avg,
    "timed_region_launches": timed_n,
    "bench_line_avg_launch_us_same_run": bench_line["roofline"]["avg_launch_us"] if bench_line else None,
    "FETCH_SIZE_KiB_per_launch": step_f,
    "WRITE_SIZE_KiB_per_launch": step_w,
    "raw_bytes_per_launch": raw,
    "calibration": ratios,
    "corrected_read_bytes_per_launch": corr_r,
    "corrected_write_bytes_per_launch": corr_w,
    "hbm_bytes_per_launch": corr_r + corr_w,
}
json.dump(traffic, open(os.path.join(dst, f"traffic_{R}.json"), "w"), indent=1)

with open(os.path.join(dst, f"{R}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary, round {R}\n\n")
    ts, tw = META[2].split(":")[1].split("/")
    ps, pw = META[3].split(":")[1].split("/")
    f.write(f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --mode {MODE} "
            f"--chunk {CHUNK} --steps {ts} --warmup {tw}` (tools/gpu_profile.sh {R} {MODE} {CHUNK}); PMC passes: `--pmc FETCH_SIZE` "
            f"and `--pmc WRITE_SIZE` in separate runs of the same command with `--steps {ps} --warmup {pw}`, plus the same "
            "passes over tools/calib (known 64 MiB copies).\n\n")
    f.write("## Kernel stats (top rows)\n\n| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
    for r in stats[:6]:
        f.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |\n")
    f.write("\n## PMC (mean per launch, KiB)\n\n| run | kernel | launches | FETCH_SIZE | WRITE_SIZE |\n|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r['run']} | {r['kernel']} | {r['launches']} | {r['FETCH_SIZE_KiB']} | {r['WRITE_SIZE_KiB']} |\n")
    f.write("\n## Calibration on known 64 MiB copies\n\n| kernel | FETCH ratio | WRITE ratio |\n|---|---|---|\n")
    for k, v in ratios.items():
        f.write(f"| {k} | {v['fetch_ratio']:.3f} | {v['write_ratio']:.3f} |\n")
    if timed_avg is not None:
        f.write(f"\nTimed region (last {timed_n} dispatches of the step kernel in the per-dispatch trace): **{timed_avg / 1e3:.2f} us/launch**"
                + (f"; the bench line printed by the same run reports avg_launch_us = {bench_line['roofline']['avg_launch_us']:.2f} "
                   f"(HIP events), value = {bench_line['value']:.4g} env-steps/s, placement = {bench_line['config'].get('placement')}.\n"
                   if bench_line else ".\n")
                + "(The all-dispatch average in the table above also covers the placement-tuning, spin-up and warm-up launches.)\n")
    f.write(f"\nStep kernel: raw {(raw) / 1e6:.1f} MB/launch; corrected read {corr_r / 1e6:.1f} MB + write {corr_w / 1e6:.1f} MB = "
            f"**{(corr_r + corr_w) / 1e6:.1f} MB/launch** at {float(step_stat['AverageNs']) / 1e3:.2f} us/launch "
            f"= {(corr_r + corr_w) / float(step_stat['AverageNs']):.0f} GB/s of real traffic.\n")
print(open(os.path.join(dst, f"{R}_summary.md")).read())
