#!/usr/bin/env python3
"""Host-side cost per rollout chunk of the strong-scaling job's per-GPU share (2^17 CartPole envs, 256-step chunks, gather of
the final tensors per chunk): can one Python process keep a GPU busy when a chunk is only ~0.22 ms of kernel time?

    python tools/chunk_overhead.py [--n 131072] [--chunks 400] [--comm torch|mxv]

World size 1 (gpurun exposes one GPU), but the gather goes through the real transport: torch.distributed (nccl backend, a
1-rank group: same Python + RCCL launch path as at 8 ranks, minus the wire time) or the C ABI's mxv_allgather_outputs.
Prints wall time per chunk vs kernel time per chunk."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 17)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--chunks", type=int, default=400)
    ap.add_argument("--comm", default="torch")
    ap.add_argument("--normal-priority", action="store_true", help="RCCL's kernels on a normal-priority stream (TORCH_NCCL_HIGH_PRIORITY=0)")
    ap.add_argument("--skip-wait", action="store_true",
                    help="MEASUREMENT ONLY (unsafe): do not order the engine's stream after the gather that last read a snapshot set — "
                         "what does that stream-wait packet cost per chunk?")
    args = ap.parse_args()
    os.environ["TORCH_NCCL_HIGH_PRIORITY"] = "0" if args.normal_priority else "1"
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from gym_amd.distributed import ShardedRollout

    sr = ShardedRollout("CartPole-v1", args.n, rank=0, world_size=1, device=0, seed=0, action_seed=1, comm=args.comm)
    sr.reset(seed=0)
    traj = sr.engine.trajectory_buffers(args.chunk)
    sr._force_collective = True      # the real collective call at world size 1 (ShardedRollout otherwise short-cuts it to a local copy)
    gather = sr.gather_async
    if args.skip_wait:
        sr._wait_set = lambda idx: None

    def run(chunks, with_gather):
        for _ in range(chunks):
            sr.rollout_per_step(args.chunk, out=traj)
            if with_gather:
                gather()

    out = {"n": args.n, "chunk": args.chunk, "comm": args.comm, "skip_wait": bool(args.skip_wait), "rccl_high_priority": not args.normal_priority}
    for with_gather in (False, True):
        run(50, with_gather)
        sr.synchronize(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(sr.engine.stream)
        run(args.chunks, with_gather)
        e1.record(sr.engine.stream)
        t_issue = time.perf_counter() - t0          # host time to ISSUE everything (no synchronisation inside)
        sr.synchronize(); torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        key = "gather" if with_gather else "rollout_only"
        out[key] = {"wall_us_per_chunk": wall / args.chunks * 1e6, "host_issue_us_per_chunk": t_issue / args.chunks * 1e6,
                    "stream_us_per_chunk": e0.elapsed_time(e1) / args.chunks * 1e3}
    print(json.dumps(out), flush=True)
    sr.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
