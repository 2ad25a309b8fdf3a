#!/usr/bin/env python3
"""bench.py — env-steps/s of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: CartPole-v1, num_envs = 2^20 (BASELINE.json's metric: the 2^20 logical envs are
partitioned over the N GPUs, --scaling strong, the default; --scaling weak keeps 2^20 envs PER GPU), on-device autoreset,
Philox-sampled actions, inputs/state resident in HBM.  A "step" is one vector step (SyncVectorEnv.step_wait) of
every env of the job; EVERY step writes its observations, rewards, terminated/truncated flags and the sampled
actions to its own slice of [chunk][N] trajectory tensors in HBM (nothing is skipped or overwritten in cache).
--mode fused (default) runs a chunk of --chunk steps as ONE kernel launch with the env state in registers;
--mode graph / eager launch the same kernel once per step.  At N > 1 every rank steps its shard of the logical
envs with no data-path collective; the final obs/reward/terminated/truncated tensors of each --chunk steps are
all-gathered over RCCL asynchronously (north_star: all-gather only for the final tensors).

Timing.  The timed region is --steps vector steps, bracketed by barrier + synchronize.  When that is shorter than
--min-timed-ms (a 20-step region is 0.12 ms: below the resolution of a host fence and of the clock ramp) the region is
REPEATED back to back inside the same bracket, `config.repeats` times `steps` steps, and every reported figure is
per step of that longer run (`ms_per_step` = bracket / (repeats * steps), `value` = envs * repeats * steps / bracket).
The launch shape does not depend on --steps: the rollout always advances in --chunk-step launches.

Rank 0 prints ONE JSON line.  `roofline` prices the step kernel against HBM: achieved = algorithmic bytes per
launch (see algorithmic_bytes_per_env_step) / average launch duration measured with HIP events on
the engine's stream over the timed region.  `cpu_baseline` (N=1 only) times the C port of the reference
(oracle/, kind "port") on one host core over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

# the host driver supports dmabuf IPC only: RCCL / cross-process device memory needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_TOTAL = 1 << 20      # BASELINE.json metric: num_envs = 2^20
ENV_ID = "CartPole-v1"
S_DIM, O_DIM = 4, 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_env_step(mode: str, chunk: float) -> float:
    """SURVEY.md §8(d).  One launch per step (eager/graph): read {state, action, counter} + write {state, obs,
    reward, 2 flags, counter} at the fp32 contract = 8*S + 4*O + 4 + 4 + 2 + 8 = 66 B for CartPole.  Fused chunk of
    K steps with the state resident in registers: outputs only, 4*O + 4 + 4 + 2, plus the state round trip
    amortised over the chunk, 16*S/K — K being the steps the timed launches REALLY fused (bench passes the measured
    steps per launch, not the --chunk argument)."""
    if mode == "fused":
        return 4 * O_DIM + 4 + 4 + 2 + 16.0 * S_DIM / chunk
    return 8 * S_DIM + 4 * O_DIM + 4 + 4 + 2 + 8


def timed_repeats(steps: int, chunk: int, local_envs: int, min_timed_ms: float, mode: str = "fused") -> int:
    """How often the `steps`-step timed region is repeated inside one bracket: a pure function of the arguments (every rank must
    issue the same launches and collectives), from a nominal 6 us per 2^20-env step, rounded up so that repeats * steps is a
    whole number of chunk-step launches (20 steps x 512 = 40 launches of 256)."""
    import math

    nominal_ms_per_step = 6.0e-3 * local_envs / ENVS_TOTAL
    repeats = max(1, math.ceil(min_timed_ms / max(steps * nominal_ms_per_step, 1e-9)))
    if repeats > 1 and mode == "fused":
        unit = chunk // math.gcd(steps, chunk)
        repeats = -(-repeats // unit) * unit
    return repeats


def cpu_baseline(sample_steps: int):
    """C port of the reference (oracle/classic_control.c, kind "port") on the host cores, same workload, bounded sample:
    first one thread (2^20 envs x sample_steps steps), then one thread per host core over equal env shards (the port has no
    cross-env dependency, exactly like one SyncVectorEnv process per core for the reference).  `value` is the
    all-core aggregate, `cores` the threads used; the single-thread rate is reported beside it."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle.oracle import OracleVecEnv

    n = ENVS_TOTAL

    def run(envs, steps, offset=0):
        o = OracleVecEnv(0, envs, 500, seed=0, action_seed=1, env_offset=offset)
        o.reset(seed=0)
        o.rollout(2)
        t0 = time.perf_counter()
        o.rollout(steps)
        return time.perf_counter() - t0

    dt1 = run(n, sample_steps)
    single = n * sample_steps / dt1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 64))
    shard = max(4, (n // cores) // 4 * 4)
    steps_all = max(sample_steps, int(2.0 * single / shard))  # ~2 s of wall time if every thread runs at the single-thread rate
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL: the shards really run concurrently
        list(ex.map(lambda i: run(shard, steps_all, i * shard), range(cores)))
    dt_all = time.perf_counter() - t0
    value = shard * cores * steps_all / dt_all
    return {
        "value": max(value, single),
        "unit": "env-steps/s",
        "cores": cores if value >= single else 1,
        "kind": "port",
        "single_core_value": single,
        "reference_python": {"value": 8.0e4, "unit": "env-steps/s/core", "kind": "reference",
                             "note": "gym.vector.SyncVectorEnv(CartPole-v1) itself, measured in the build container (BASELINE.md §2); "
                                     "/root/reference does not exist on the GPU box, so it cannot be re-timed beside this line — the C port "
                                     "above is the reference's arithmetic without the Python interpreter"},
        "sample": f"{ENV_ID}, Philox actions + autoreset, gcc -O2 C port of the reference's step loop: 1 thread, 2^20 envs x "
                  f"{sample_steps} steps ({dt1:.1f} s); {cores} threads x {shard} envs x {steps_all} steps ({dt_all:.1f} s incl. "
                  "thread start-up and resets); the Python reference itself measured 8.0e4 env-steps/s/core (BASELINE.md §2)",
    }


def measure_variant(args, ShardedRollout, torch):
    """Same workload, MXV_FLAG_REWARD_F32 | MXV_FLAG_ACTION_I32 outputs; short (a tenth of the headline's steps)."""
    sr = ShardedRollout(ENV_ID, ENVS_TOTAL, rank=0, world_size=1, device=torch.cuda.current_device(), seed=0,
                        action_seed=1, reward_f32=True, action_i32=True)
    eng = sr.engine
    sr.reset(seed=0)
    placement = None
    if args.placement_candidates > 1:
        traj, placement = eng.tuned_trajectory_buffers(args.chunk, candidates=args.placement_candidates)
    else:
        traj = eng.trajectory_buffers(args.chunk)
    launches = max(8, args.steps // args.chunk // 4)
    t_spin = time.perf_counter()  # same clock-ramp treatment as the headline: --spinup-ms of untimed work first
    while (time.perf_counter() - t_spin) * 1e3 < max(args.spinup_ms, 1.0):
        sr.rollout_per_step(args.chunk, mode="fused", out=traj, record_actions=True)
        sr.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(eng.stream)
    for _ in range(launches):
        sr.rollout_per_step(args.chunk, mode="fused", out=traj, record_actions=True)
    ev1.record(eng.stream)
    sr.synchronize()
    ms = ev0.elapsed_time(ev1) / launches
    sr.close()
    b = algorithmic_bytes_per_env_step("fused", args.chunk)
    steps_s = ENVS_TOTAL * args.chunk / (ms * 1e-3)
    return {"value": steps_s, "unit": "env-steps/s", "us_per_step": ms * 1e3 / args.chunk,
            "outputs": "float32 rewards, int32 actions (26 real B/env-step)", "roofline_frac": steps_s * b / 1e9 / HBM_PEAK_GBS,
            "placement_candidates_us_per_step": None if placement is None else placement.get("us_per_step")}


def read_traffic(mode: str, steps_per_launch: float, envs: int, compact: bool):
    """(HBM bytes per launch, source) for THIS launch shape from the committed PMC passes (profiles/traffic_*.json, written by
    tools/summarize_profile.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs; counters cannot be read from inside the
    process that is being timed), or (None, reason).  The profile's bytes per env-step are only transferable to a launch
    with the same kernel, output dtypes and steps per launch: anything else reports null instead of a mismatched number."""
    if compact:
        return None, "no PMC pass of the compact-output kernel committed"
    pdir = os.path.join(ROOT, "profiles")
    try:
        for name in sorted((f for f in os.listdir(pdir) if f.startswith("traffic_") and f.endswith(".json")), reverse=True):
            with open(os.path.join(pdir, name)) as f:
                j = json.load(f)
            if j.get("mode", "eager") != mode:
                continue
            if mode == "fused" and abs(float(j.get("chunk", 0)) - steps_per_launch) > 0.5:
                continue
            per_env_step = float(j["hbm_bytes_per_launch"]) / float(j["env_steps_per_launch"]) if "env_steps_per_launch" in j \
                else float(j["hbm_bytes_per_launch"]) / (float(j.get("chunk", 1) if mode == "fused" else 1) * float(j.get("num_envs", ENVS_TOTAL)))
            return per_env_step * envs * steps_per_launch, f"profiles/{name} (separate rocprofv3 --pmc passes of the same launch shape, scaled per env-step)"
    except Exception as e:  # noqa: BLE001
        return None, f"profiles/traffic_*.json unreadable: {e}"
    return None, "no committed PMC pass with this launch shape"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20480)
    ap.add_argument("--warmup", type=int, default=2048)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default, BASELINE.json's metric): 2^20 logical envs in total, 2^20/N per GPU; weak: 2^20 per GPU")
    ap.add_argument("--chunk", type=int, default=256,
                    help="steps per fused launch = rollout length between all-gathers of the final tensors (a typical "
                         "on-policy horizon; 9.3 GB of trajectory tensors at 2^20 envs)")
    ap.add_argument("--mode", default="fused", choices=["fused", "graph", "eager"],
                    help="fused: one launch per chunk, env state in registers; graph/eager: one launch per step")
    ap.add_argument("--no-graph", action="store_true", help="alias of --mode eager")
    ap.add_argument("--min-timed-ms", type=float, default=60.0,
                    help="the timed region is repeated back to back until it is nominally at least this long (see docstring)")
    ap.add_argument("--repeats", type=int, default=0, help="force the number of repeats of the timed region (0 = from --min-timed-ms)")
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="untimed device spin-up before the W warmup steps (DVFS: the GPU needs tens of ms of load to "
                         "reach its sustained clocks; a 2000-step run is 13 ms); 0 disables; reported in config")
    ap.add_argument("--compact-outputs", action="store_true",
                    help="float32 rewards + int32 actions (MXV_FLAG_REWARD_F32|ACTION_I32: 26 real bytes per env-step "
                         "instead of 34); off by default: the headline keeps the reference's float64 / int64 dtypes")
    ap.add_argument("--placement-candidates", type=int, default=8,
                    help="> 1: time that many candidate sets of trajectory tensors before the run and keep the fastest")
    ap.add_argument("--cpu-sample-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary compact-outputs measurement (N=1 only)")
    ap.add_argument("--comm", default="torch", choices=["torch", "mxv"],
                    help="transport of the per-chunk all-gather at N > 1: torch.distributed (packed all_gather_into_tensor) or the C "
                         "ABI's own RCCL collective (mxv_allgather_outputs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1.  nccl (= RCCL) is the product path; gloo exists to exercise the "
                         "multi-rank control flow on a box with fewer GPUs than ranks (ranks then share devices)")
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
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()   # debug path: more ranks than GPUs
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's kernels run on high-priority streams: a chunk's all-gather gets CUs as soon as rollout waves retire instead of
        # queueing behind the next chunk's (long-running, chip-filling) rollout launch
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from gym_amd.distributed import ShardedRollout

    total_envs = ENVS_TOTAL * (world if args.scaling == "weak" else 1)
    if total_envs % (4 * world):
        raise SystemExit(f"{total_envs} envs do not split into {world} shards of a multiple of 4 envs")
    local_envs = total_envs // world
    sr = ShardedRollout(ENV_ID, total_envs, rank=rank, world_size=world, device=local_rank, seed=0, action_seed=1,
                        reward_f32=args.compact_outputs, action_i32=args.compact_outputs, comm=args.comm)
    eng = sr.engine
    mode = "eager" if args.no_graph else args.mode
    sr.reset(seed=0)
    # [chunk][N] obs / reward / flags / actions, reused every chunk
    placement = None
    if mode == "fused" and args.placement_candidates > 1:
        try:
            traj, placement = eng.tuned_trajectory_buffers(args.chunk, candidates=args.placement_candidates)
        except (RuntimeError, MemoryError) as e:   # e.g. out of device memory: measure on the first allocation instead
            torch.cuda.empty_cache()
            traj, placement = eng.trajectory_buffers(args.chunk), {"error": f"placement tuning failed: {e}"[:300]}
    else:
        traj = eng.trajectory_buffers(args.chunk)
    launches = [0]
    since_gather = [0]

    def run(steps, gather=True):
        """`steps` vector steps as chunk-step launches; at N > 1 the final tensors are all-gathered (asynchronously, overlapping
        the next launch) every time --chunk steps have accumulated — the cadence does not depend on how `steps` was cut."""
        done = 0
        while done < steps:
            k = min(args.chunk, steps - done)
            sr.rollout_per_step(k, mode=mode, out=traj, record_actions=True)
            launches[0] += 1 if mode == "fused" else k
            done += k
            since_gather[0] += k
            if world > 1 and gather and since_gather[0] >= args.chunk:
                sr.gather_async()
                since_gather[0] = 0

    def fence():
        sr.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # device spin-up: same workload, untimed, until --spinup-ms of wall time has passed; then W warmup steps
    # (which also instantiate the hipGraph(s) and RCCL communicators used in the timed region)
    # The spin-up is time-based, so its iteration count differs between ranks: it must not contain a collective (every rank
    # has to issue the same sequence of them) — the all-gathers start with the warm-up, whose step count is fixed.
    spin_steps = 0
    if args.spinup_ms > 0:
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            run(args.chunk, gather=False)
            sr.synchronize()
            spin_steps += args.chunk
    fence()   # ranks leave placement tuning and spin-up at different times
    since_gather[0] = 0
    run(max(args.warmup, 1))
    if world > 1:
        sr.gather()
    fence()

    repeats = args.repeats if args.repeats > 0 else timed_repeats(args.steps, args.chunk, local_envs, args.min_timed_ms, mode)
    timed_steps = args.steps * repeats

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    fence()
    launches[0] = 0
    since_gather[0] = 0
    t0 = time.perf_counter()
    ev0.record(eng.stream)
    run(timed_steps)
    ev1.record(eng.stream)
    if world > 1:
        sr.wait_gather()
    fence()
    elapsed = time.perf_counter() - t0

    launch_ms = ev0.elapsed_time(ev1) / launches[0]  # avg step-kernel launch duration on the engine's stream
    steps_per_launch = timed_steps / launches[0]
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        value = total_envs * timed_steps / elapsed
        b_env_step = algorithmic_bytes_per_env_step(mode, steps_per_launch)
        algo_bytes = b_env_step * local_envs * steps_per_launch
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        traffic, traffic_source = read_traffic(mode, steps_per_launch, local_envs, args.compact_outputs)
        out = {
            "metric": "env-steps/sec at num_envs=2^20, CartPole-v1" if args.scaling == "strong"
                      else "env-steps/sec at num_envs=2^20 per GPU, CartPole-v1",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / timed_steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{ENV_ID}, num_envs={total_envs} ({local_envs} per GPU), on-device autoreset + "
                            "Philox4x32-10 sampled actions, fp64 state (BASELINE.json configs[1])",
                "num_envs_per_gpu": local_envs,
                "repeats": repeats,
                "timed_steps": timed_steps,
                "timed_region_ms": elapsed * 1e3,
                "launch": {"fused": f"fused: 1 launch per {args.chunk}-step chunk, env state in registers",
                           "graph": "1 launch per step, hipGraph replay", "eager": "1 launch per step, eager"}[mode],
                "outputs": "per-step obs/reward/terminated/truncated/actions written to [chunk][N] trajectory tensors"
                           + (" (float32 rewards, int32 actions)" if args.compact_outputs else
                              " (float64 rewards, int64 actions: the reference's dtypes)"),
                "chunk": args.chunk,
                "placement": placement if placement is not None else "first allocation (no placement tuning)",
                "spinup": f"{spin_steps} untimed steps ({args.spinup_ms:.0f} ms) before the {args.warmup} warmup steps (clock ramp)",
                "parallelism": f"env-shard x{world}" + (f", async RCCL all-gather of the final tensors every {args.chunk} steps "
                                                        f"({args.comm} transport)" if world > 1 else ""),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "rollout_kernel_v3<CartPole>" if mode == "fused" else "step_kernel<CartPole>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_env_step": b_env_step,
                "algorithmic_bytes_per_launch": algo_bytes,
                "env_steps_per_launch": local_envs * steps_per_launch,
                "steps_per_launch": steps_per_launch,
                "avg_launch_us": launch_ms * 1e3,
            },
        }
        if mode == "fused" and not args.compact_outputs and local_envs % 1024 == 0:
            # what THIS box sustains for the kernel's store pattern with the physics removed (mxv_write_probe, include/mxv.h), into
            # the very tensors the timed region wrote: boxes of this pool differ by 20 % here, so the kernel's time
            # is printed next to the box's own ceiling for it
            from gym_amd import _native
            torch.cuda.synchronize()
            probe_us = _native.write_probe(local_rank, local_envs, args.chunk, 20, traj["obs"], traj["reward"], traj["actions"],
                                           traj["terminated"], traj["truncated"])
            real_b = 34.0 * local_envs
            out["roofline"]["write_probe"] = {
                "what": "same store pattern, no physics (mxv_write_probe), same tensors",
                "us_per_step": probe_us, "real_GBs": real_b / probe_us / 1e3,
                "kernel_us_per_step": launch_ms * 1e3 / steps_per_launch,
                "kernel_over_probe": launch_ms * 1e3 / steps_per_launch / probe_us}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_steps)

    sr.close()
    if rank == 0:
        if world == 1 and mode == "fused" and not args.compact_outputs and not args.no_variants:
            # reported beside the headline, never instead of it: the same workload with the engine's contract-minimal output
            # dtypes (float32 rewards, int32 actions: SURVEY.md §8d's algorithmic bytes, 26 real B/env-step instead of 34)
            out["variants"] = {"compact_outputs": measure_variant(args, ShardedRollout, torch)}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
