#!/usr/bin/env python3
"""bench.py — env-steps/s of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: CartPole-v1, num_envs = 2^20 PER GPU (weak scaling), on-device autoreset,
Philox-sampled actions, inputs/state resident in HBM.  A "step" is one vector step of every env of the job =
one launch of the step kernel per GPU.  At N > 1 every rank steps its shard of the 2^20*N logical envs with no
data-path collective; the final obs/reward/terminated/truncated tensors of each chunk of --chunk steps are
all-gathered over RCCL asynchronously (north_star: all-gather only for the final tensors).

Rank 0 prints ONE JSON line.  `roofline` prices the step kernel against HBM: achieved = algorithmic bytes per
launch (66 B per CartPole env-step) / average launch duration measured with HIP events on the
engine's stream over the timed region.  `cpu_baseline` (N=1 only) times the C port of the reference
(oracle/, kind "port") on one host core over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 1 << 20
ENV_ID = "CartPole-v1"
ALGO_BYTES_PER_ENV_STEP = 66  # 8*S + 4*O + 4 + 4 + 2 + 8 with S=4, O=4
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(sample_steps: int):
    """C port of the reference (oracle/classic_control.c), one thread, same workload, bounded sample."""
    from oracle.oracle import OracleVecEnv

    n = ENVS_PER_GPU
    o = OracleVecEnv(0, n, 500, seed=0, action_seed=1)
    o.reset(seed=0)
    o.rollout(2)
    t0 = time.perf_counter()
    o.rollout(sample_steps)
    dt = time.perf_counter() - t0
    return {
        "value": n * sample_steps / dt,
        "unit": "env-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{ENV_ID}, 2^20 envs x {sample_steps} steps, Philox actions + autoreset, gcc -O2 C port of the "
                  f"reference's step loop ({dt:.1f} s); the Python reference itself measured 8.0e4 env-steps/s/core "
                  "(BASELINE.md §2)",
    }


def read_traffic():
    """HBM bytes per launch from the committed PMC passes (profiles/traffic_*.json), or None."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted(f for f in os.listdir(pdir) if f.startswith("traffic_") and f.endswith(".json"))
        if not names:
            return None
        with open(os.path.join(pdir, names[-1])) as f:
            return float(json.load(f)["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--chunk", type=int, default=100, help="steps per launch batch / per all-gather")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--cpu-sample-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs a launcher: python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; gym_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from gym_amd.distributed import ShardedRollout

    total_envs = ENVS_PER_GPU * world
    sr = ShardedRollout(ENV_ID, total_envs, rank=rank, world_size=world, device=local_rank, seed=0, action_seed=1,
                        reward_f32=False)
    eng = sr.engine
    use_graph = not args.no_graph
    sr.reset(seed=0)

    def run(steps):
        done = 0
        while done < steps:
            k = min(args.chunk, steps - done)
            sr.rollout(k, use_graph=use_graph)
            if world > 1:
                sr.gather_async()
            done += k

    def fence():
        sr.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # warmup (also instantiates the hipGraph(s) and RCCL communicators used in the timed region)
    run(args.warmup)
    if args.steps % args.chunk:
        sr.rollout(args.steps % args.chunk, use_graph=use_graph)
    if world > 1:
        sr.gather()
    fence()

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record(eng.stream)
    run(args.steps)
    ev1.record(eng.stream)
    fence()
    elapsed = time.perf_counter() - t0

    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # avg step-kernel launch duration on the engine's stream
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        value = total_envs * args.steps / elapsed
        algo_bytes = ALGO_BYTES_PER_ENV_STEP * ENVS_PER_GPU
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec at num_envs=2^20 per GPU, CartPole-v1",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{ENV_ID}, num_envs=2^20 per GPU ({total_envs} total), on-device autoreset + "
                            "Philox4x32-10 sampled actions, fp64 state (BASELINE.json configs[1])",
                "num_envs_per_gpu": ENVS_PER_GPU,
                "launch": "hipGraph" if use_graph else "eager",
                "chunk": args.chunk,
                "parallelism": f"env-shard x{world}" + (", async RCCL all-gather of final tensors per chunk" if world > 1 else ""),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "step_kernel<CartPole>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": read_traffic(),
                "algorithmic_bytes_per_launch": algo_bytes,
                "avg_launch_us": kernel_ms * 1e3,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_steps)
        print(json.dumps(out), flush=True)

    sr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
