#!/usr/bin/env python3
"""What does NormalizeObservation / NormalizeReward over a [K = 128][2^20] CartPole trajectory cost, kernel by kernel?  Run under
    rocprofv3 --kernel-trace --stats -d gpurun_out/<tag> -- python tools/norm_ab.py
(the bench's measure_normalize, three repetitions); prints the event-timed totals, the kernel split is in the trace's stats."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.normalize import measure_normalize  # noqa: E402

for rep in range(3):
    r = measure_normalize(torch, 1 << 20, 128)
    print(json.dumps({k: {"us_per_step": round(r[k]["us_per_step"], 3), "frac": round(r[k]["roofline"]["frac"], 3)} for k in ("normalize_obs", "normalize_reward")}), flush=True)
