#!/usr/bin/env python3
"""What `out = r.rollout_per_step(K)` costs per call in a learner's loop (no `out=`): the first calls sort their trajectory tensors by
HBM class (probe launches, parked allocations), later calls get blocks back from torch's caching allocator that gym_amd.placement
remembers — wall time per call, probe report and the rollout's own time."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--calls", type=int, default=8)
    args = ap.parse_args()
    import torch
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout("CartPole-v1", args.n, seed=0, action_seed=1)
    r.reset(seed=0)
    print(json.dumps({"num_device_free_stat": torch.cuda.memory_stats(r.device).get("num_device_free")}))
    out = None
    for i in range(args.calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = r.rollout_per_step(args.chunk)       # the previous set is released by this assignment, AFTER the call
        r.synchronize()
        dt = time.perf_counter() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(r.stream)
        r.rollout_per_step(args.chunk, out=out)
        e1.record(r.stream)
        r.synchronize()
        rep = dict(getattr(r, "last_placement", None) or {})
        print(json.dumps({"call": i, "wall_ms": round(dt * 1e3, 2), "rollout_us_per_step": round(e0.elapsed_time(e1) * 1e3 / args.chunk, 3),
                          "placement": {k: rep.get(k) for k in ("balanced", "remembered", "candidates", "parked_GiB", "seconds")}}), flush=True)
    r.close()


if __name__ == "__main__":
    main()
