#!/usr/bin/env python3
"""tools/ipc_probe.py — does cross-process device memory (hipIpcGetMemHandle / hipIpcOpenMemHandle: what RCCL's intra-node transports and
torch's CUDA-tensor sharing are built on) work on this box with HSA_ENABLE_IPC_MODE_LEGACY = 0 / 1 / unset?  bench.py and
tools/config_bench_dist.py set the variable to 0 before the HIP runtime starts (the task environment says the host driver supports dmabuf IPC
only); this is the evidence for it on the box at hand.  One JSON line per setting: a child process opens the parent's device tensor and sums it.
    python tools/ipc_probe.py            # runs the three settings as sub-processes
"""
import json
import os
import subprocess
import sys


def _child(conn):
    import torch

    t = conn.recv()
    conn.send(float(t.sum().item()))


def one():
    import torch
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    p = ctx.Process(target=_child, args=(b,))
    p.start()
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    a.send(x)
    ok, detail = False, ""
    if a.poll(60):
        got = a.recv()
        ok = abs(got - float(x.sum().item())) < 1.0
        detail = f"child summed {got:.0f}"
    p.join(30)
    if not ok:
        detail = detail or f"child exit code {p.exitcode}, no answer"
    print(json.dumps({"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "ipc_ok": ok, "detail": detail}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for v in ("0", "1", None):
            env = {k: w for k, w in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
            if v is not None:
                env["HSA_ENABLE_IPC_MODE_LEGACY"] = v
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True, timeout=90)
            except subprocess.TimeoutExpired:
                print(json.dumps({"HSA_ENABLE_IPC_MODE_LEGACY": v, "ipc_ok": False, "detail": "hung: no answer within 90 s, killed"}), flush=True)
                continue
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if line:
                print(line[-1], flush=True)
            else:
                err = [l for l in p.stderr.splitlines() if "rror" in l][-2:]
                print(json.dumps({"HSA_ENABLE_IPC_MODE_LEGACY": v, "ipc_ok": False, "detail": f"rc {p.returncode}: " + " | ".join(err)[:300]}), flush=True)
