#!/usr/bin/env python3
"""bench.py — env-steps/s of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # also fine: an external launcher is detected through WORLD_SIZE

Workload = BASELINE.json configs[1]: CartPole-v1, num_envs = 2^20 (partitioned over the N GPUs: --scaling strong, the default;
--scaling weak keeps 2^20 envs PER GPU), on-device autoreset, Philox-sampled actions, state resident in HBM.  A "step" is one
vector step (SyncVectorEnv.step_wait, gym/vector/sync_vector_env.py:135-169) of every env of the job; EVERY step writes its
observations, rewards, terminated/truncated flags and sampled actions to its own slice of [chunk][N] trajectory tensors in HBM.
--mode fused (default) runs --chunk steps as ONE kernel launch with the env state in registers.  At N > 1 every rank steps its
shard with no data-path collective; the final obs/reward/terminated/truncated tensors of every --gather-every steps are
all-gathered over RCCL asynchronously, overlapping the next launches (north_star; gym/vector/utils/numpy_utils.py:49-50).

Timing.  The timed region is --steps vector steps, bracketed by barrier + synchronize.  When that is shorter than --min-timed-ms
the region is REPEATED back to back inside the same bracket, `config.repeats` times, and every figure is per step of that longer
run.  The launch shape does not depend on --steps: the rollout always advances in --chunk-step launches.

Output.  stdout carries ONE JSON line of less than 4 KB (tests/test_gpu_bench_line.py): the contract fields, `config`
(workload, launch shape, work_check, launch_info, per-rank reports), `roofline` (HIP events on the engine's stream; algorithmic
bytes per SURVEY.md §8d), `cpu_baseline` (N = 1: the C port of the reference on the host cores, bounded sample; the Python
reference's committed figure beside it) and `variants` = {group: [us_per_step, roofline_frac]}.  The measurement itself lives in benchmarks/headline.py (this
file: arguments + the CPU baseline, the only code of the bench that may touch oracle/).  The full record of every
secondary measurement (benchmarks/variants.py) goes to stderr, one JSON object per group, and to
gpurun_out/bench_variants.json (--variants-file); the long-form copy of the headline goes to gpurun_out/bench_headline.json.
"""
import argparse
import json
import os
import sys
import time

# The host driver of this pool supports dmabuf IPC only (task environment: the variable is exported on the GPU boxes; without it RCCL /
# cross-process device memory fails with `hipIpcGetMemHandle: invalid argument`): set before the HIP runtime starts, never overridden.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchmarks.common import (CHECK_ENVS, ENV_ID, ENVS_TOTAL, HBM_PEAK_GBS, algorithmic_bytes_per_env_step, read_traffic,  # noqa: E402,F401
                               read_valu, spinup_steps, timed_repeats, warm_until_stable, work_checksum)
from benchmarks.headline import LINE_LIMIT, compact_line, run, self_launch  # noqa: E402,F401


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baselines (the ONLY place the bench touches oracle/: the checker timed as the host baseline, never the thing measured)
# ---------------------------------------------------------------------------------------------------------------------------
def reference_python_baseline():
    """The reference itself: gym.vector.SyncVectorEnv(CartPole-v1) under tools/reference_baseline.py.  The committed
    run (profiles/reference_cpu_baseline.json: build container, host described in the file) is always reported; where the
    reference tree exists (the build container — not the GPU box) the headline case is re-timed live beside it."""
    out = {"kind": "reference", "unit": "env-steps/s/core"}
    path = os.path.join(ROOT, "profiles", "reference_cpu_baseline.json")
    try:
        with open(path) as f:
            j = json.load(f)
        out.update({"value": j["headline"]["value"], "env_id": j["headline"]["env_id"], "num_envs": j["headline"]["num_envs"],
                    "source": "profiles/reference_cpu_baseline.json (tools/reference_baseline.py, measured " + j.get("measured_at", "?") + ")",
                    "host": j.get("host"), "single_core_all_sizes": j["single_core"].get("CartPole-v1"),
                    "all_cores": j.get("all_cores")})
    except (OSError, KeyError, ValueError) as e:
        out.update({"value": None, "source": f"profiles/reference_cpu_baseline.json unreadable: {e}"})
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import reference_baseline as rb

        if rb.reference_available():
            live = rb.measure(quick=True)
            out["live"] = {"value": live["headline"]["value"], "host": live["host"],
                           "what": "re-timed in this process's host (CartPole-v1, n = 64, 1000 steps, best of 1)"}
        else:
            out["live"] = None   # no /root/reference here (GPU box): the committed run stands
    except Exception as e:  # noqa: BLE001
        out["live"] = {"error": str(e)[:200]}
    return out


def cpu_baseline(sample_steps: int):
    """C port of the reference (oracle/classic_control.c, kind "port") on the host cores, same workload, bounded sample:
    first one thread (2^20 envs x sample_steps steps), then one thread per host core over equal env shards (the port has no
    cross-env dependency, exactly like one SyncVectorEnv process per core for the reference).  `value` is the
    all-core aggregate, `cores` the threads used; the single-thread rate is reported beside it, and so is the Python
    reference itself (reference_python_baseline)."""
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
        "reference_python": reference_python_baseline(),
        "sample_short": f"C port (gcc -O2): 1 thread 2^20 envs x {sample_steps} steps {dt1:.1f} s; {cores} threads x {shard} envs x {steps_all} steps {dt_all:.1f} s",
        "sample": f"{ENV_ID}, Philox actions + autoreset, gcc -O2 C port of the reference's step loop: 1 thread, 2^20 envs x "
                  f"{sample_steps} steps ({dt1:.1f} s); {cores} threads x {shard} envs x {steps_all} steps ({dt_all:.1f} s incl. "
                  "thread start-up and resets)",
    }


# ---------------------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------------------
def parse_args():
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
                    help="nominal length of the untimed device spin-up before the W warmup steps (DVFS: the GPU needs tens of ms of "
                         "load to reach its sustained clocks); converted to a fixed number of steps; 0 disables; reported in config")
    ap.add_argument("--warm-max-s", type=float, default=4.0,
                    help="upper bound of the rate-settling phase that precedes reset(seed=0) (an idle box needs > 1 s of load); 0 disables")
    ap.add_argument("--compact-outputs", action="store_true",
                    help="float32 rewards + int32 actions (MXV_FLAG_REWARD_F32|ACTION_I32: 26 real bytes per env-step "
                         "instead of 34); off by default: the headline keeps the reference's float64 / int64 dtypes")
    ap.add_argument("--placement", default="sorted", choices=["sorted", "placed", "first", "off"],
                    help="trajectory tensors: sorted = ordinary allocations sorted by measured HBM class (the product default, "
                         "DeviceRollout.trajectory_buffers); placed = 256-MiB physical chunks of measured class mapped through the HIP "
                         "virtual-memory API (mxv_placed_alloc); "
                         "first = the first ordinary allocation; off = first, and MXV_PLACEMENT=off for every measurement of the run "
                         "(no probe launch, no memory parked anywhere: the setting that cannot fail)")
    ap.add_argument("--cpu-sample-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurements (N=1 only)")
    ap.add_argument("--comm", default="torch", choices=["torch", "mxv"],
                    help="transport of the per-chunk all-gather at N > 1: torch.distributed (packed all_gather_into_tensor) or the C "
                         "ABI's own RCCL collective (mxv_allgather_outputs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1.  nccl (= RCCL) is the product path; gloo exists to exercise the "
                         "multi-rank control flow on a box with fewer GPUs than ranks (ranks then share devices)")
    ap.add_argument("--gather-every", type=int, default=1024,
                    help="steps between all-gathers of the final tensors at N > 1 = the rollout horizon whose final obs / reward / done the "
                         "learner receives (a multiple of --chunk: the horizon advances in chunk-step launches).  xGMI is per-link bound: "
                         "an 8-rank all-gather of the 3.4-MB final tensors costs about as much as 256 steps of a 2^17-env shard (0.18 ms), so "
                         "the horizon, not the launch, sets the cadence — fewer, larger collectives")
    ap.add_argument("--force-gather", action="store_true",
                    help="issue the per-chunk all-gather at N = 1 too (with --comm mxv: a real one-rank RCCL communicator and "
                         "ncclAllGather per output tensor on the side stream) — the gather's launch path measured on one GPU")
    ap.add_argument("--variants-file", default=os.path.join(ROOT, "gpurun_out", "bench_variants.json"),
                    help="where the full record of the secondary measurements goes (the stdout line carries one pair per group); '' disables")
    ap.add_argument("--headline-file", default=os.path.join(ROOT, "gpurun_out", "bench_headline.json"),
                    help="long-form copy of the headline record (prose, per-rank placement reports); '' disables")
    ap.add_argument("--init-timeout", type=float, default=180.0, help="seconds the process group / first collective may take")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds a self-launched job may take in total")
    return ap.parse_args()




def main():
    run(parse_args(), __file__, cpu_baseline)


if __name__ == "__main__":
    main()
