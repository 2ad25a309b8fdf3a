#!/usr/bin/env python3
"""Resources and K-step-loop instruction counts of the trajectory rollout kernels (one per env kind, OUT = 1 and 2), from a
cross-compile of gym_amd/csrc/mxv_kernels.hip (no GPU needed).
    python tools/loop_stats.py [workdir] [-- extra hipcc flags]
Prints one JSON line per kernel: VGPRs, SGPRs, scratch, occupancy, and for the K-step loop: instructions, VALU, f64 VALU, branches."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
work = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "--" else "/tmp/mxv_loop_stats"
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
os.makedirs(work, exist_ok=True)
src = os.path.join(ROOT, "gym_amd", "csrc", "mxv_kernels.hip")
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-c", src,
                    "-o", os.path.join(work, "k.o"), "-Rpass-analysis=kernel-resource-usage", "-save-temps"] + extra, cwd=work,
                   capture_output=True, text=True)
if p.returncode:
    sys.exit(p.stderr[-3000:])
import test_kernel_resources as tk   # noqa: E402  (the ISA helpers of the resource test)

asm = open(os.path.join(work, [f for f in os.listdir(work) if f.endswith("gfx950.s")][0])).read()
res = tk._resources(p.stderr)
NAMES = ["CartPole", "Pendulum", "Acrobot", "MountainCar", "MountainCarContinuous"]
EPL = {0: 2, 1: 1, 2: 1, 3: 2, 4: 2}
for env in range(5):
    for out in (1, 2):
        sym = f"_ZN3mxv12_GLOBAL__N_117rollout_kernel_v3ILi{env}ELb1ELi{EPL[env]}ELb0ELi{out}ELb0ELi0EEEvNS_8StepArgsE"
        if sym not in res:
            continue
        body = tk._function_body(asm, sym)
        loops = [(h, t) for h, t in tk._inner_loops(body) if "global_store" in t]
        text = max(loops, key=lambda ht: ht[1].count("global_store"))[1]
        ins = [l.strip().split()[0] for l in text.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        r = res[sym]
        print(json.dumps({"kernel": f"{NAMES[env]} E={EPL[env]} OUT={out}", "VGPRs": r.get("VGPRs"), "SGPRs": r.get("TotalSGPRs"),
                          "scratch": r.get("ScratchSize"), "occupancy": r.get("Occupancy"), "spill": r.get("VGPRs Spill"),
                          "loop_instructions": len(ins), "loop_valu": sum(1 for i in ins if i.startswith("v_")),
                          "loop_valu_f64": sum(1 for i in ins if i.startswith("v_") and "f64" in i),
                          "loop_salu": sum(1 for i in ins if i.startswith("s_") and not i.startswith(("s_cbranch", "s_waitcnt", "s_nop"))),
                          "loop_branches": sum(1 for i in ins if i.startswith("s_cbranch")), "loop_stores": sum(1 for i in ins if i.startswith("global_store")),
                          "loop_lds": sum(1 for i in ins if i.startswith("ds_")), "loop_waitcnt": sum(1 for i in ins if i.startswith("s_waitcnt")),
                          "loop_v_mov": sum(1 for i in ins if i.startswith("v_mov")), "loop_cndmask": sum(1 for i in ins if i.startswith("v_cndmask"))}))
