#!/usr/bin/env python3
"""BASELINE.json configs[3] and configs[4] on N GPUs of one node (bench.py reports configs[1] only):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/config_bench_dist.py

  config 4: Acrobot-v1, 2^19 envs per GPU (2^22 on 8), fused rollout chunks, RCCL all-gather of the final tensors per chunk
  config 5: {CartPole, Pendulum, Acrobot, MountainCar} x 2^15 envs per GPU (2^20 on 8), four streams per GPU, gather per chunk
Weak scaling like bench.py: per-GPU work fixed, value = total env-steps / max-over-ranks time.  One JSON line per config on
rank 0.  Runs with WORLD_SIZE=1 too (the gather degenerates to a local copy)."""
import argparse
import json
import os
import sys
import time

# the host driver supports dmabuf IPC only: RCCL / cross-process device memory needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MXV_DIST_BACKEND", "nccl")   # "gloo": exercise the multi-rank control flow on fewer GPUs than ranks
    if backend == "gloo":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    from gym_amd.distributed import ShardedRollout
    from gym_amd.mixed import DEFAULT_MIX, MixedRollout

    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=256, help="steps per fused launch = steps per all-gather")
    ap.add_argument("--steps", type=int, default=4096)
    a = ap.parse_args()
    chunk, steps, warm = a.chunk, a.steps // a.chunk * a.chunk, 2 * a.chunk

    def fence(obj):
        obj.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(obj, traj, total_envs, label, extra):
        def run(n, gather=True):
            for _ in range(n // chunk):
                obj.rollout_per_step(chunk, out=traj)
                if gather:
                    obj.gather_async()
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.15:   # clock ramp; time-based, so no collective inside (counts differ per rank)
            run(chunk, gather=False)
            obj.synchronize()
        fence(obj)
        run(warm)
        obj.gather()
        fence(obj)
        t0 = time.perf_counter()
        run(steps)
        fence(obj)
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"config": label, "n_gpus": world, "total_envs": total_envs, "steps": steps, "chunk": chunk,
                              "us_per_step": round(float(dt) / steps * 1e6, 3),
                              "env_steps_per_s": float(f"{total_envs * steps / float(dt):.4g}"), **extra}), flush=True)

    sr = ShardedRollout("Acrobot-v1", (1 << 19) * world, rank=rank, world_size=world, device=local, seed=0, action_seed=1)
    sr.reset(seed=0)
    traj = sr.engine.trajectory_buffers(chunk)
    rep = getattr(sr.engine, "last_placement", None) or {}
    timed(sr, traj, (1 << 19) * world, "config4: Acrobot-v1, 2^19 envs per GPU, all-gather of final tensors per chunk",
          {"placement": {k: rep.get(k) for k in ("mode", "balanced", "parked_GiB")}})
    sr.close()
    del traj

    single = bool(os.environ.get("MXV_MIXED_SINGLE_LAUNCH"))   # A/B hook: all segments in ONE launch (mxv_rollout_mixed) instead of one launch per segment on its own stream
    mr = MixedRollout((1 << 17) * world, DEFAULT_MIX, rank=rank, world_size=world, device=local, seed=0, action_seed=1,
                      single_launch=single)
    mr.reset(seed=0)
    timed(mr, mr.trajectory_buffers(chunk), (1 << 17) * world,
          "config5: " + "+".join(DEFAULT_MIX) + ", 2^15 envs of each kind per GPU, "
          + ("ONE launch (block -> segment table)" if single else "4 launches on 4 streams") + ", gather per chunk", {})
    mr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
