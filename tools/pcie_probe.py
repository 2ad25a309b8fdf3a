#!/usr/bin/env python3
"""PCIe bandwidth of the GPU box (pinned and pageable, both directions, one and two concurrent streams) — the ceiling of the
gym-compatible NumPy loop (HipVectorEnv.step moves actions in and obs / reward / flags out every step)."""
import json
import time

import numpy as np
import torch

dev = torch.device("cuda", 0)
out = {}
for mb in (1, 8, 32, 128):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    hp = torch.empty(n, dtype=torch.uint8).pin_memory()
    hg = torch.empty(n, dtype=torch.uint8)
    for name, src, dst in (("h2d_pinned", hp, d), ("d2h_pinned", d, hp), ("h2d_pageable", hg, d), ("d2h_pageable", d, hg)):
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[f"{name}_{mb}MB_GBs"] = round(n / dt / 1e9, 2)
    # both directions at once on two streams
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d2 = torch.empty(n, dtype=torch.uint8, device=dev)
    hp2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        with torch.cuda.stream(s1):
            d.copy_(hp, non_blocking=True)
        with torch.cuda.stream(s2):
            hp2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize()
    out[f"bidir_pinned_{mb}MB_GBs_each"] = round(n / ((time.perf_counter() - t0) / 10) / 1e9, 2)
# host memcpy speed (what copy=True costs on top)
a = np.empty(32 << 20, np.uint8); b = np.empty_like(a)
t0 = time.perf_counter()
for _ in range(10):
    np.copyto(b, a)
out["host_memcpy_32MB_GBs"] = round(a.nbytes / ((time.perf_counter() - t0) / 10) / 1e9, 2)
print(json.dumps(out))
