import json, sys, torch
sys.path.insert(0, '.')
from benchmarks.toy_text import measure_tabular
for rep in range(2):
    for gid in ("FrozenLake8x8-v1", "Taxi-v3", "CliffWalking-v0"):
        for compact in (False, True):
            for gen in (True, False):
                r = measure_tabular(torch, gid, 1 << 20, 128, compact=compact, general_kernel=gen)
                print(json.dumps({"gid": gid, "compact": compact, "general": gen, "us_per_step": round(r["us_per_step"], 3), "stored_GBs": round(r["stored_GBs"]), "kernel": r["kernel"], "balanced": (r.get("placement") or {}).get("balanced")}), flush=True)
