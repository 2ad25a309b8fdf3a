#!/usr/bin/env python3
"""bench.py — env-steps/s of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # also fine: an external launcher is detected through WORLD_SIZE

Workload = BASELINE.json configs[1]: CartPole-v1, num_envs = 2^20 (BASELINE.json's metric: the 2^20 logical envs are
partitioned over the N GPUs, --scaling strong, the default; --scaling weak keeps 2^20 envs PER GPU), on-device autoreset,
Philox-sampled actions, inputs/state resident in HBM.  A "step" is one vector step (SyncVectorEnv.step_wait,
gym/vector/sync_vector_env.py:135-169) of every env of the job; EVERY step writes its observations, rewards,
terminated/truncated flags and the sampled actions to its own slice of [chunk][N] trajectory tensors in HBM (nothing is skipped
or overwritten in cache).  --mode fused (default) runs a chunk of --chunk steps as ONE kernel launch with the env state in
registers; --mode graph / eager launch the same kernel once per step.  At N > 1 every rank steps its shard of the logical envs
with no data-path collective; the final obs/reward/terminated/truncated tensors of every --gather-every steps (the rollout horizon: 1024,
four launches) are all-gathered over RCCL asynchronously, overlapping the next launches (north_star: all-gather only for the final tensors;
the reference's np.stack, gym/vector/utils/numpy_utils.py:49-50).

Timing.  The timed region is --steps vector steps, bracketed by barrier + synchronize.  When that is shorter than
--min-timed-ms (a 20-step region is 0.12 ms: below the resolution of a host fence and of the clock ramp) the region is
REPEATED back to back inside the same bracket, `config.repeats` times `steps` steps, and every reported figure is
per step of that longer run (`ms_per_step` = bracket / (repeats * steps), `value` = envs * repeats * steps / bracket).
The launch shape does not depend on --steps: the rollout always advances in --chunk-step launches.

Rank 0 prints ONE JSON line.  `roofline` prices the step kernel against HBM: achieved = algorithmic bytes per
launch (see algorithmic_bytes_per_env_step) / average launch duration measured with HIP events on
the engine's stream over the timed region.  `cpu_baseline` (N=1 only) times the C port of the reference
(oracle/, kind "port") on the host cores over a bounded sample of the same workload, and carries the reference itself
(gym.vector.SyncVectorEnv, timed by tools/reference_baseline.py: the committed run, plus a live re-timing where
/root/reference exists).  `config.work_check` lets a reader hold the timed region against the oracle
(tests/test_gpu_bench_line.py does).  `variants` (N=1): the other BASELINE.json configs on one GPU and the
learner-in-the-loop step(actions) path, each with its own roofline.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

# The host driver of this pool supports dmabuf IPC only (task environment: the variable is exported on the GPU boxes; without it RCCL /
# cross-process device memory fails with `hipIpcGetMemHandle: invalid argument`): set before the HIP runtime starts, never overridden.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_TOTAL = 1 << 20      # BASELINE.json metric: num_envs = 2^20
ENV_ID = "CartPole-v1"
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4   # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at the nominal 2.4 GHz
DIMS = {"CartPole-v1": (4, 4), "Pendulum-v1": (2, 3), "Acrobot-v1": (4, 6), "MountainCar-v0": (2, 2), "MountainCarContinuous-v0": (2, 2)}  # (S, O)
CHECK_ENVS = 4096         # work_check: the first CHECK_ENVS envs of rank 0's shard


def algorithmic_bytes_per_env_step(mode: str, chunk: float, env_id: str = ENV_ID) -> float:
    """SURVEY.md §8(d).  One launch per step (eager/graph/given actions): read {state, action, counter} + write {state, obs,
    reward, 2 flags, counter} at the fp32 contract = 8*S + 4*O + 4 + 4 + 2 + 8 = 66 B for CartPole.  Fused chunk of
    K steps with the state resident in registers: outputs only, 4*O + 4 + 4 + 2, plus the state round trip
    amortised over the chunk, 16*S/K — K being the steps the timed launches REALLY fused (bench passes the measured
    steps per launch, not the --chunk argument)."""
    S, O = DIMS[env_id]
    if mode == "fused":
        return 4 * O + 4 + 4 + 2 + 16.0 * S / chunk
    return 8 * S + 4 * O + 4 + 4 + 2 + 8


def timed_repeats(steps: int, chunk: int, local_envs: int, min_timed_ms: float, mode: str = "fused") -> int:
    """How often the `steps`-step timed region is repeated inside one bracket: a pure function of the arguments (every rank must
    issue the same launches and collectives), from a nominal 6 us per 2^20-env step, rounded up so that repeats * steps is a
    whole number of chunk-step launches (20 steps x 512 = 40 launches of 256)."""
    nominal_ms_per_step = 6.0e-3 * local_envs / ENVS_TOTAL
    repeats = max(1, math.ceil(min_timed_ms / max(steps * nominal_ms_per_step, 1e-9)))
    if repeats > 1 and mode == "fused":
        unit = chunk // math.gcd(steps, chunk)
        repeats = -(-repeats // unit) * unit
    return repeats


def spinup_steps(spinup_ms: float, chunk: int, local_envs: int) -> int:
    """Untimed steps before the warm-up (clock ramp), a pure function of the arguments — a whole number of chunks worth about
    spinup_ms at the nominal 6 us per 2^20-env step — so that the step index of the timed region, and with it
    config.work_check, is reproducible."""
    if spinup_ms <= 0:
        return 0
    nominal_ms_per_chunk = 6.0e-3 * max(local_envs, 1 << 17) / ENVS_TOTAL * chunk
    return max(1, math.ceil(spinup_ms / nominal_ms_per_chunk)) * chunk


def work_checksum(terminated, truncated, actions):
    """64-bit checksum of a [K][n] block of flags and discrete actions: sum over (k, i) of (terminated + 2 truncated + 4 action) *
    ((k * n + i) * 0x9E3779B97F4A7C15 + 1) mod 2^64.  Works on torch tensors (device) and NumPy arrays (the oracle side of
    tests/test_gpu_bench_line.py) alike: int64 / uint64 arithmetic wraps."""
    import numpy as np

    if isinstance(terminated, np.ndarray):
        K, n = terminated.shape
        v = terminated.astype(np.uint64) + np.uint64(2) * truncated.astype(np.uint64) + np.uint64(4) * actions.astype(np.uint64)
        idx = np.arange(K * n, dtype=np.uint64).reshape(K, n)
        with np.errstate(over="ignore"):
            w = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
            return int((v * w).sum(dtype=np.uint64))
    import torch

    K, n = terminated.shape
    v = terminated.to(torch.int64) + 2 * truncated.to(torch.int64) + 4 * actions.to(torch.int64)
    idx = torch.arange(K * n, dtype=torch.int64, device=terminated.device).reshape(K, n)
    w = idx * (0x9E3779B97F4A7C15 - (1 << 64)) + 1      # the same constant as a wrapped int64
    return int((v * w).sum().item()) % (1 << 64)


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baselines
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
        "sample": f"{ENV_ID}, Philox actions + autoreset, gcc -O2 C port of the reference's step loop: 1 thread, 2^20 envs x "
                  f"{sample_steps} steps ({dt1:.1f} s); {cores} threads x {shard} envs x {steps_all} steps ({dt_all:.1f} s incl. "
                  "thread start-up and resets)",
    }


# ---------------------------------------------------------------------------------------------------------------------------
# secondary measurements (N = 1)
# ---------------------------------------------------------------------------------------------------------------------------
def _spin(fn, sync, ms):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        fn()
        sync()


def warm_until_stable(fn, sync, max_s=4.0, window_s=0.1, tol=0.01, min_s=0.6):
    """Run fn until its rate has settled: successive windows of window_s agree within tol (and at least min_s have passed), or max_s.
    A box that has been idle needs more than a second of load before its clocks, and with them the kernel's instruction stream, reach
    their sustained state (the first process on a fresh box measured 6.4-6.5 us per step after 0.2 s of load, 5.8 after 3 s;
    profiles/r3d_*).  Returns (seconds, calls)."""
    t_start = time.perf_counter()
    prev, calls = None, 0
    while True:
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < window_s:
            fn()
            sync()
            n += 1
        calls += n
        now = time.perf_counter()
        rate = n / (now - t0)
        if (prev is not None and abs(rate - prev) <= tol * rate and now - t_start >= min_s) or now - t_start >= max_s:
            return now - t_start, calls
        prev = rate


def measure_fused(torch, env_id, envs, chunk, *, compact=False, launches=8, spin_ms=60.0, probe=True, valu=False):
    """One env kind, fused trajectory launches on one GPU: us per step, env-steps/s, roofline on the algorithmic bytes, the
    write probe of its own store pattern into the same tensors."""
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout(env_id, envs, seed=0, action_seed=1, reward_f32=compact, action_i32=compact)
    r.reset(seed=0)
    traj = r.trajectory_buffers(chunk)
    placement = getattr(r, "last_placement", None) if sum(t.numel() * t.element_size() for t in traj.values()) >= _native.SORTED_MIN_BYTES else None
    fn = lambda: r.rollout_per_step(chunk, out=traj)   # noqa: E731
    _spin(fn, r.stream.synchronize, spin_ms)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(r.stream)
    for _ in range(launches):
        fn()
    ev1.record(r.stream)
    r.synchronize()
    us = ev0.elapsed_time(ev1) / launches / chunk * 1e3
    b = algorithmic_bytes_per_env_step("fused", chunk, env_id)
    real_b = sum(t[0].numel() * t.element_size() for k, t in traj.items()) / envs
    out = {"workload": f"{env_id}, num_envs={envs}, fused {chunk}-step launches, "
                       + ("float32 rewards + int32 actions" if compact else "the reference's output dtypes"),
           "value": envs / us * 1e6, "unit": "env-steps/s", "us_per_step": us,
           "roofline": {"bound": "hbm", "algorithmic_bytes_per_env_step": b, "achieved": envs * b / us / 1e3, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": envs * b / us / 1e3 / HBM_PEAK_GBS, "stored_bytes_per_env_step": real_b}}
    out["launch_info"] = r.handle.last_launch()
    if env_id == ENV_ID and envs == ENVS_TOTAL:      # the headline configuration: the committed PMC pass of this launch shape, if any
        tr, src = read_traffic("fused", chunk, envs, compact)
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr, src
        if tr:
            out["roofline"]["traffic_over_algorithmic"] = tr / (b * envs * chunk)
    if valu:
        per_env_step, source = read_valu(env_id)     # wave64 VALU instructions a wave issues per env-step of each of its lanes
        if per_env_step:
            rate = envs / us * 1e6 * per_env_step / 64.0
            out["roofline_valu"] = {"bound": "valu", "valu_instructions_per_env_step": per_env_step, "achieved": rate,
                                    "peak": VALU_PEAK_WAVE_INSTR_PER_S, "unit": "wave64 VALU instructions/s",
                                    "frac": rate / VALU_PEAK_WAVE_INSTR_PER_S, "source": source}
        else:
            out["roofline_valu"] = {"bound": "valu", "frac": None, "source": source}
    if probe:
        torch.cuda.synchronize()
        flags = (_native.FLAG_REWARD_F32 | _native.FLAG_ACTION_I32) if compact else 0
        p = _native.write_probe_env(r.device.index, r.spec.kind, flags, envs, chunk, 8, traj["obs"], traj["reward"], traj["actions"],
                                    traj["terminated"], traj["truncated"])
        out["write_probe"] = {"us_per_step": p, "stored_GBs": real_b * envs / p / 1e3, "kernel_over_probe": us / p}
    if placement is not None:
        out["placement"] = placement
    r.close()
    del traj
    torch.cuda.empty_cache()
    return out


def measure_mixed(torch, envs_per_segment, chunk, launches=8, spin_ms=60.0):
    """BASELINE.json configs[4]'s per-GPU share: {CartPole, Pendulum, Acrobot, MountainCar} x envs_per_segment, four streams."""
    from gym_amd.mixed import DEFAULT_MIX, MixedRollout

    total = envs_per_segment * len(DEFAULT_MIX)
    mr = MixedRollout(total, rank=0, world_size=1, seed=0, action_seed=1)
    mr.reset(seed=0)
    fn = lambda: mr.rollout(chunk)   # noqa: E731
    _spin(fn, mr.synchronize, spin_ms)
    t0 = time.perf_counter()
    for _ in range(launches):
        fn()
    mr.synchronize()
    us = (time.perf_counter() - t0) / launches / chunk * 1e6
    b = sum(algorithmic_bytes_per_env_step("fused", chunk, e) for e in DEFAULT_MIX) / len(DEFAULT_MIX)
    mr.close()
    return {"workload": f"mixed batch {list(DEFAULT_MIX)} x {envs_per_segment} envs each (configs[4]'s share of one of 8 GPUs), one stream per "
                        f"segment, fused {chunk}-step launches, final tensors only",
            "value": total / us * 1e6, "unit": "env-steps/s", "us_per_step": us,
            "roofline": {"bound": "latency (512 single-wave Acrobot workgroups on 1024 SIMDs set the floor, DESIGN.md §4)",
                         "algorithmic_bytes_per_env_step": b, "achieved": total * b / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": total * b / us / 1e3 / HBM_PEAK_GBS}}


def _hbm(us, envs, bytes_per_env_step, **extra):
    gbs = envs * bytes_per_env_step / us / 1e3
    return dict({"value": envs / us * 1e6, "unit": "env-steps/s", "us_per_step": us,
                 "roofline": {"bound": "hbm", "algorithmic_bytes_per_env_step": bytes_per_env_step, "achieved": gbs, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}}, **extra)


def _event_us(torch, stream, fn, reps, steps_per_call):
    fn()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / reps / steps_per_call * 1e3


def measure_normalize(torch, envs, chunk, reps=6):
    """NormalizeObservation / NormalizeReward (gym/wrappers/normalize.py:57-144) on the [K][N] trajectory tensors of a
    fused CartPole rollout: per chunk, the batch moments of every step (one streaming read), then the affine map with the statistics as
    they stood after that step's update (read + write).  Algorithmic bytes per env-step: observations 4 O (sums) + 4 O + 4 O (apply,
    float32 out) = 48; rewards 8 + 2 (sums: reward + both flags) + 8 + 8 (apply) = 26."""
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    dr = DeviceRollout(ENV_ID, envs, seed=0, action_seed=1)
    dr.reset(seed=0)
    tr = dr.rollout_per_step(chunk, out=dr.trajectory_buffers(chunk, layout="separate"))
    dr.synchronize()
    s, O = dr.stream, dr.O
    no, nr = _native.Norm(O, envs, stream=s.cuda_stream), _native.Norm(1, envs, stream=s.cuda_stream)
    with torch.cuda.stream(s):
        y32 = torch.empty((chunk, envs, O), dtype=torch.float32, device=dr.device)
        o64 = torch.empty((chunk, envs), dtype=torch.float64, device=dr.device)
    out = {"workload": f"{ENV_ID}, num_envs={envs}, the [K={chunk}][N] trajectory tensors of one fused launch normalised in place of the "
                       "reference's per-step wrappers (running mean / var updated once per step, exactly their order)",
           "normalize_obs": _hbm(_event_us(torch, s, lambda: no.observations(chunk, tr["obs"], y32, True, 1e-8), reps, chunk), envs, 12 * O,
                                 kernels="mxv_norm.hip: obs sums (read 4 O) + scan + apply (read 4 O, write 4 O float32)"),
           "normalize_reward": _hbm(_event_us(torch, s, lambda: nr.rewards(chunk, tr["reward"], False, tr["terminated"], tr["truncated"], o64, 0.99, 1e-8),
                                              reps, chunk), envs, 26, kernels="mxv_norm.hip: discounted-return sums (read 8 + 2) + scan + apply (read 8, write 8)")}
    # the batch moments formed by the rollout itself (mxv_set_obs_partials): what NormalizeObservation then costs ON TOP of the rollout
    try:
        tr = None                                                  # (the set normalised above: its numbers are taken, its 9 GiB are needed)
        torch.cuda.empty_cache()
        trp = dr.trajectory_buffers(chunk, obs_partials=True)     # sorted by HBM class, as a caller gets them by default
        plain = {k: t for k, t in trp.items() if k != "obs_partials"}
        nf = _native.Norm(O, envs, stream=s.cuda_stream)
        r0 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=plain), reps, chunk)
        r1 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=trp), reps, chunk)
        sums = torch.empty((chunk, 2 * O), dtype=torch.float64, device=dr.device)

        def fused():
            nf.obs_sums_partials(chunk, trp["obs_partials"], trp["obs_partials"].shape[1], sums)
            nf.obs_apply(chunk, trp["obs"], y32, True, 1e-8, sums.unsqueeze(0), 1, envs)

        nfu = _event_us(torch, s, fused, reps, chunk)
        inc = nfu + (r1 - r0)
        b = 8 * O + 2 * 16 * O * trp["obs_partials"].shape[1] / envs   # apply: read 4 O + write 4 O; partials: 2 O doubles per tile, written + read
        out["normalize_obs_fused_moments"] = dict(_hbm(inc, envs, b, kernels="rollout_kernel_v3<..., STATS> writes per-tile column sums; mxv_norm.hip: tree over the "
                                                                              "partials + scan + apply (read 4 O, write 4 O float32); no pass reads the observations back"),
                                                  rollout_us_per_step=r0, rollout_with_partials_us_per_step=r1, normalize_from_partials_us_per_step=nfu,
                                                  separate_us_per_step=out["normalize_obs"]["us_per_step"],
                                                  note="us_per_step = what normalisation adds to the rollout: (rollout with partials - rollout) + tree + scan + apply")
        # ... and NormalizeReward's discounted returns (mxv_set_return_partials), alone and together with the observation moments
        nrf = _native.Norm(1, envs, stream=s.cuda_stream)

        class _Returns:      # what DeviceRollout.fuse_reward_normalizer needs of a normaliser: its returns array and its discount
            gamma = 0.99
            backend = nrf

        dr.fuse_reward_normalizer(_Returns)
        leaves = trp["obs_partials"].shape[1]
        with torch.cuda.stream(s):
            rp = torch.empty((chunk, leaves, 2), dtype=torch.float64, device=dr.device)
        rets, both = dict(plain, ret_partials=rp), dict(trp, ret_partials=rp)
        r2 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=rets), reps, chunk)
        r3 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=both), reps, chunk)
        rsums = torch.empty((chunk, 2), dtype=torch.float64, device=dr.device)

        def fused_reward():
            nrf.reward_sums_partials(chunk, rp, leaves, rsums)
            nrf.reward_apply(chunk, trp["reward"], False, o64, 1e-8, rsums.unsqueeze(0), 1, envs)

        nru = _event_us(torch, s, fused_reward, reps, chunk)
        out["normalize_reward_fused_moments"] = dict(_hbm(nru + (r2 - r0), envs, 16 + 2 * 16 * leaves / envs,
                                                          kernels="rollout_kernel_v3<..., STATS = 2> advances the discounted returns; tree + scan + apply (read 8, write 8)"),
                                                     rollout_with_partials_us_per_step=r2, normalize_from_partials_us_per_step=nru,
                                                     separate_us_per_step=out["normalize_reward"]["us_per_step"])
        out["rollout_and_both_normalisations"] = {"separate_us_per_step": r0 + out["normalize_obs"]["us_per_step"] + out["normalize_reward"]["us_per_step"],
                                                  "fused_us_per_step": r3 + nfu + nru, "rollout_with_both_partials_us_per_step": r3}
        dr.handle.set_obs_partials(None)
        dr.handle.set_return_partials(None, 0.0, None)            # nothing may point into nrf's returns any more
        nf.close(), nrf.close()
        del trp, plain, sums, rp, rets, both, rsums
    except Exception as e:  # noqa: BLE001
        out.setdefault("normalize_obs_fused_moments", {"error": f"{type(e).__name__}: {e}"[:300]})
        out.setdefault("normalize_reward_fused_moments", {"error": f"{type(e).__name__}: {e}"[:300]})
    no.close(), nr.close(), dr.close()
    del tr, y32, o64
    torch.cuda.empty_cache()
    return out


def measure_tabular(torch, gid, envs, chunk, reps=6, compact=False, general_kernel=False):
    """a toy_text env as a table-driven kernel (gym/envs/toy_text/frozen_lake.py:247-270, taxi.py:270-278): fused K-step
    rollouts with sampled actions, every step's obs / actions (int64), reward / prob (float64) and both flags written to [K][N]
    trajectory tensors.  Contract bytes as SURVEY.md §8(d) prices them (4-byte scalars, 1-byte flags): obs 4 + action 4 + reward 4 +
    prob 4 + 2 = 18; stored with the reference's dtypes: 34."""
    from gym_amd.toy_text import TabularRollout

    r = TabularRollout(gid, envs, seed=0, action_seed=1, compact=compact, general_kernel=general_kernel)
    r.reset(seed=0)
    out = r.trajectory_buffers(chunk)
    us = _event_us(torch, r.stream, lambda: r.rollout_per_step(chunk, out=out), reps, chunk)
    stored = 18 if compact else 34
    res = _hbm(us, envs, 18, workload=f"{gid}, num_envs={envs}, fused {chunk}-step launches, "
                                      + ("int32 obs / actions + float32 reward / prob" if compact else "the reference's dtypes") + f" ({stored} B stored per env-step)",
               stored_GBs=envs * stored / us / 1e3, placement=getattr(r, "last_placement", None),
               kernel={1: "tab_step_kernel (general)", 2: "tab_traj_kernel (integer thresholds, packed table)"}.get(r.handle.last_kernel()))
    r.close()
    del out
    torch.cuda.empty_cache()
    return res


def measure_blackjack(torch, envs, chunk, reps=6):
    """Blackjack-v1 (gym/envs/toy_text/blackjack.py:108-160): fused K-step rollouts, observation = three int64 columns, reward float64,
    flags, sampled actions int64 -> 42 B stored per env-step; contract bytes (4-byte scalars): 3 x 4 + 4 + 4 + 2 = 22."""
    from gym_amd import _native

    dev = torch.device("cuda", torch.cuda.current_device())
    h = _native.Blackjack(envs, seed=0, action_seed=1)
    obs = torch.empty((chunk, 3, envs), dtype=torch.int64, device=dev)
    rew = torch.empty((chunk, envs), dtype=torch.float64, device=dev)
    term, trunc = (torch.empty((chunk, envs), dtype=torch.uint8, device=dev) for _ in range(2))
    act = torch.empty((chunk, envs), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    h.reset()
    run = lambda: h.rollout(chunk, obs, rew, term, trunc, None, actions_out_dev=act, per_step=True)   # noqa: E731
    for _ in range(2):
        run()
    h.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    h.sync()
    us = (time.perf_counter() - t0) / reps / chunk * 1e6
    res = _hbm(us, envs, 22, workload=f"Blackjack-v1, num_envs={envs}, fused {chunk}-step launches, the reference's dtypes (42 B stored per env-step)",
               stored_GBs=envs * 42 / us / 1e3, episodes_ended_per_env_step=float(((term | trunc) != 0).float().mean().item()))
    h.close()
    del obs, rew, term, trunc, act
    torch.cuda.empty_cache()
    return res


def measure_numpy_loop(envs, steps):
    """SURVEY.md §8(d), the third number: the gym-compatible loop — gym_amd.make(id, num_envs) stepped with NumPy actions, NumPy
    observations / rewards / flags / infos coming back (gym/vector/sync_vector_env.py:135-169 as a caller sees it) — PCIe and Python
    inclusive.  This is what a user who swaps gym.vector.SyncVectorEnv for the engine and changes nothing else gets; it is never `value`."""
    import numpy as np

    import gym_amd

    env = gym_amd.make(ENV_ID, num_envs=envs)
    env.reset(seed=0)
    env.action_space.seed(0)
    acts = [env.action_space.sample() for _ in range(4)]
    for i in range(6):
        env.step(acts[i % 4])
    t0 = time.perf_counter()
    for i in range(steps):
        out = env.step(acts[i % 4])        # the caller's loop and nothing else (a `term.sum()` per step here cost 0.5 ms at 2^20 envs:
    us = (time.perf_counter() - t0) / steps * 1e6   # NumPy's bool -> int64 reduction, more than half of what was being measured)
    ended = 0
    for i in range(8):                     # untimed: that episodes end and autoreset on this path too
        _, _, term, trunc, _ = env.step(acts[i % 4])
        ended += int(np.count_nonzero(term)) + int(np.count_nonzero(trunc))
    del out
    env.close()
    return {"workload": f"{ENV_ID}, num_envs={envs}, gym_amd.make(...).step(actions) with NumPy arrays in and out (copy=True, infos with "
                        "final_observation), host loop", "us_per_step": us, "value": envs / us * 1e6, "unit": "env-steps/s",
            "bytes_over_pcie_per_env_step": 8 + 16 + 8 + 2, "pcie_GBs": envs * 34 / us / 1e3, "episodes_ended": ended, "episodes_ended_over": "8 untimed steps after the loop",
            "note": "PCIe- and Python-inclusive; never the bench value"}


def measure_step_loop(torch, envs, steps=600, compact=False, halves=1):
    """The learner-in-the-loop path: DeviceRollout.step(actions) with caller-provided actions, one launch per vector step
    (gym/vector/sync_vector_env.py:131-169 with a policy in the loop).  halves = 2: the batch as two half-size engines (global env
    indices unchanged: env_offset) on their own streams, stepped alternately — the double-buffered sampling pattern (the policy
    works on one half while the other steps) that lets one half's launch overlap the other's tail."""
    from gym_amd.rollout import DeviceRollout

    n = envs // halves
    eng = [DeviceRollout(ENV_ID, n, env_offset=i * n, seed=0, action_seed=1, reward_f32=compact, action_i32=compact) for i in range(halves)]
    acts = []
    for e in eng:
        e.reset(seed=0)
        with torch.cuda.stream(e.stream):
            acts.append(e.sample_actions().clone())
        e.synchronize()

    def one():
        for e, a in zip(eng, acts):
            with torch.cuda.stream(e.stream):      # the caller works on the engine's stream: no cross-stream wait per step
                e.step(a, want_final=False)

    _spin(one, lambda: [e.stream.synchronize() for e in eng], 60.0)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in eng]
    for e, (a0, _) in zip(eng, evs):
        a0.record(e.stream)
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    for e, (_, a1) in zip(eng, evs):
        a1.record(e.stream)
    for e in eng:
        e.synchronize()
    wall_us = (time.perf_counter() - t0) / steps * 1e6
    gpu_us = max(a0.elapsed_time(a1) for a0, a1 in evs) / steps * 1e3
    for e in eng:
        e.close()
    b = algorithmic_bytes_per_env_step("given", 1)
    us = max(wall_us, gpu_us)
    return {"halves": halves, "dtypes": "float32 rewards, int32 actions" if compact else "float64 rewards, int64 actions (the reference's)",
            "us_per_step": us, "gpu_us_per_step": gpu_us, "value": envs / us * 1e6, "unit": "env-steps/s",
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_env_step": b, "achieved": envs * b / us / 1e3, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": envs * b / us / 1e3 / HBM_PEAK_GBS}}


def measure_policy_loop(torch, envs, per_graph=32, steps=1920):
    """The learner-in-the-loop path where launches, not kernels, bound it: CartPole-v1, `envs` envs (a PPO-sized batch), a linear
    policy's three kernels between the steps.  (i) the loop as a caller writes it: one ctypes call and three torch ops per step;
    (ii) the same loop recorded ONCE into a hipGraph of the caller's — DeviceRollout.enable_graph_capture() moves the step index into
    device memory, so replays continue the streams (tests/test_gpu_graph_capture.py: == single calls, bit for bit) — and replayed."""
    from gym_amd.rollout import DeviceRollout

    def loop(captured):
        r = DeviceRollout(ENV_ID, envs, seed=0, action_seed=1)
        r.reset(seed=0)
        torch.manual_seed(0)
        W = torch.randn(r.O, 2, device=r.device)

        def one():
            r.step((r.obs @ W).argmax(dim=1), want_final=False)

        with torch.cuda.stream(r.stream):
            for _ in range(64):
                one()
            r.stream.synchronize()
            if captured:
                r.enable_graph_capture()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=r.stream):
                    for _ in range(per_graph):
                        one()
                run, calls = g.replay, steps // per_graph
            else:
                run, calls = one, steps
            for _ in range(max(2, calls // 8)):
                run()
            r.stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                run()
            r.stream.synchronize()
            us = (time.perf_counter() - t0) / steps * 1e6
        ended = int(r.handle.get_episodes().sum())
        r.close()
        return us, ended

    eager, e1 = loop(False)
    graph, e2 = loop(True)
    return {"workload": f"{ENV_ID}, num_envs={envs}, obs @ W -> argmax -> step(actions), host wall time per vector step",
            "one_call_per_step": {"us_per_step": eager, "value": envs / eager * 1e6, "unit": "env-steps/s"},
            "recorded_in_a_hipgraph": {"steps_per_graph": per_graph, "us_per_step": graph, "value": envs / graph * 1e6, "unit": "env-steps/s"},
            "speedup": eager / graph, "episodes_ended": [e1, e2]}


def measure_step_kernel(torch, envs, launches=400, compact=False):
    """The step kernel itself (HIP events around back-to-back launches are dominated by the inter-launch gap, so the kernel time
    is taken with one event pair PER launch on a few launches and the minimum-gap figure is the loop's)."""
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout(ENV_ID, envs, seed=0, action_seed=1, reward_f32=compact, action_i32=compact)
    r.reset(seed=0)
    with torch.cuda.stream(r.stream):
        a = r.sample_actions().clone()
        for _ in range(200):
            r.step(a, want_final=False)
        r.stream.synchronize()
        ts = []
        for _ in range(launches):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(r.stream)
            r.step(a, want_final=False)
            e1.record(r.stream)
            ts.append((e0, e1))
        r.stream.synchronize()
    us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ts)
    r.close()
    med = us[len(us) // 2]
    b = algorithmic_bytes_per_env_step("given", 1)
    return {"us_per_launch_median": med, "us_per_launch_min": us[0],
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_env_step": b, "achieved": envs * b / med / 1e3, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": envs * b / med / 1e3 / HBM_PEAK_GBS},
            "note": "event pair around single launches (includes the events' own ~1-2 us); rocprofv3 kernel time in profiles/"}


def read_valu(env_id: str):
    """(wave64 VALU instructions per wave-step of the fused trajectory kernel, source) from the latest committed PMC pass
    (profiles/valu_*.json, written by tools/gpu_valu.sh from `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES` of the kernel itself), or
    (None, reason): a counter cannot be read from inside the process being timed."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        for name in sorted((f for f in os.listdir(pdir) if f.startswith("valu_") and f.endswith(".json")), reverse=True):
            with open(os.path.join(pdir, name)) as f:
                j = json.load(f)
            if env_id in j.get("kinds", {}):
                k = j["kinds"][env_id]
                return float(k["valu_per_wave_step"]) / float(k["envs_per_lane"]), f"profiles/{name} ({k.get('kernel', 'rollout_kernel_v3')}: SQ_INSTS_VALU / (waves x steps))"
    except Exception as e:  # noqa: BLE001
        return None, f"profiles/valu_*.json unreadable: {e}"
    return None, "no committed SQ_INSTS_VALU pass for this env kind"


def read_traffic(mode: str, steps_per_launch: float, envs: int, compact: bool):
    """(HBM bytes per launch, source) for THIS launch shape from the committed PMC passes (profiles/traffic_*.json, written by
    tools/summarize_profile.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs; counters cannot be read from inside the
    process that is being timed), or (None, reason).  The profile's bytes per env-step are only transferable to a launch
    with the same kernel, output dtypes and steps per launch: anything else reports null instead of a mismatched number."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        for name in sorted((f for f in os.listdir(pdir) if f.startswith("traffic_") and f.endswith(".json")), reverse=True):
            with open(os.path.join(pdir, name)) as f:
                j = json.load(f)
            if bool(j.get("compact_outputs", False)) != bool(compact):      # float32 + int32 outputs: profiles/traffic_compact_*.json
                continue
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


# ---------------------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment as torch.distributed.run would set them), wait, and fail fast and readably if
    any rank dies or the job exceeds --launch-timeout.  Rank 0 prints the JSON line on the inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MXV_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    deadline = time.time() + args.launch_timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = next((c for c in codes if c), 0)
            if rc:
                print(f"bench.py: ranks exited with codes {codes}", file=sys.stderr, flush=True)
            break
        bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.time() > deadline:
            why = f"rank {bad[0][0]} exited with code {bad[0][1]}" if bad else f"no result after {args.launch_timeout:.0f} s"
            print(f"bench.py: {why}; stopping the other ranks", file=sys.stderr, flush=True)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = bad[0][1] if bad else 124
            break
        time.sleep(0.05)
    return rc


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
    ap.add_argument("--placement", default="sorted", choices=["sorted", "placed", "tuned", "first", "off"],
                    help="trajectory tensors: sorted = ordinary allocations sorted by measured HBM class (the product default, "
                         "DeviceRollout.trajectory_buffers); placed = 256-MiB physical chunks of measured class mapped through the HIP "
                         "virtual-memory API (mxv_placed_alloc); tuned = round 2's timing of --placement-candidates ordinary sets; "
                         "first = the first ordinary allocation; off = first, and MXV_PLACEMENT=off for every measurement of the run "
                         "(no probe launch, no memory parked anywhere: the setting that cannot fail)")
    ap.add_argument("--placement-candidates", type=int, default=8, help="candidate sets of --placement tuned")
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
    ap.add_argument("--init-timeout", type=float, default=180.0, help="seconds the process group / first collective may take")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds a self-launched job may take in total")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.placement == "off":
        os.environ["MXV_PLACEMENT"] = "off"      # inherited by self-launched ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries ONE JSON line and nothing else: RCCL ("Hostname : ... / Librccl path : ...") and gloo ("[Gloo] Rank 0 is connected
    # ...") print from C++ straight to file descriptor 1 when a communicator comes up.  The descriptor is pointed at stderr for the whole
    # run (in every rank), and the line goes out through a private duplicate of the original one.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; gym_amd has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local_rank %= ndev   # debug path: more ranks than GPUs
    elif local_rank >= ndev:
        raise SystemExit(f"bench.py: rank {rank} needs device {local_rank} but only {ndev} HIP device(s) are visible "
                         f"(--gpus {args.gpus} with --backend nccl is one process per GPU; --backend gloo shares devices)")
    torch.cuda.set_device(local_rank)
    comm_info = {"backend": None}
    solo_group = world == 1 and args.force_gather and args.comm == "torch"    # a REAL one-rank process group: torch's RCCL path, minus the links
    if solo_group and "MASTER_PORT" not in os.environ:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or solo_group:
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's kernels run on high-priority streams: a chunk's all-gather gets CUs as soon as rollout waves retire instead of
        # queueing behind the next chunk's (long-running, chip-filling) rollout launch
        from gym_amd.distributed import prefer_high_priority_collectives
        prefer_high_priority_collectives()        # TORCH_NCCL_HIGH_PRIORITY=1 unless the caller set it: explicit, this process only
        to = datetime.timedelta(seconds=args.init_timeout)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=to)
            else:
                dist.init_process_group("gloo", timeout=to)
            # first collective: how many ranks does the communicator really span?
            one = torch.ones(1, dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(one)
            ranks_seen = int(one.item())
        except Exception as e:  # noqa: BLE001
            raise SystemExit(f"bench.py: rank {rank}/{world}: process group ({args.backend}, {os.environ.get('MASTER_ADDR')}:"
                             f"{os.environ.get('MASTER_PORT')}) failed within {args.init_timeout:.0f} s: {e}")
        comm_info = {"backend": args.backend, "ranks_seen": ranks_seen, "transport": args.comm,
                     "launcher": "bench.py" if os.environ.get("MXV_BENCH_SELF_LAUNCHED") else "external (WORLD_SIZE was set)"}
        if args.backend == "nccl":
            try:
                comm_info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                comm_info["rccl_version"] = None
        if ranks_seen != world:
            raise SystemExit(f"bench.py: the communicator spans {ranks_seen} ranks, expected {world}")

    t_start = time.perf_counter()

    def trace(what):
        """Phase markers on stderr at N > 1 (never stdout: that carries the one JSON line): if a multi-GPU run stalls or a rank is slow,
        the log says where.  MXV_BENCH_TRACE=0 silences them, =1 forces them at N = 1."""
        flag = os.environ.get("MXV_BENCH_TRACE")
        if flag == "0" or (world == 1 and flag != "1"):
            return
        print(f"[bench rank {rank}/{world} +{time.perf_counter() - t_start:7.2f}s] {what}", file=sys.stderr, flush=True)

    trace(f"process group up ({comm_info.get('backend')}, ranks_seen={comm_info.get('ranks_seen', 1)}), device {local_rank}")
    from gym_amd.distributed import ShardedRollout

    total_envs = ENVS_TOTAL * (world if args.scaling == "weak" else 1)
    if total_envs % (4 * world):
        raise SystemExit(f"{total_envs} envs do not split into {world} shards of a multiple of 4 envs")
    local_envs = total_envs // world
    sr = ShardedRollout(ENV_ID, total_envs, rank=rank, world_size=world, device=local_rank, seed=0, action_seed=1,
                        reward_f32=args.compact_outputs, action_i32=args.compact_outputs, comm=args.comm)
    if solo_group:
        sr._force_collective = True     # dist.all_gather_into_tensor for real (ShardedRollout short-cuts a one-rank gather to a local copy)
    eng = sr.engine
    mode = "eager" if args.no_graph else args.mode
    sr.reset(seed=0)
    # [chunk][N] obs / reward / flags / actions, reused every chunk
    placement = None
    if mode == "fused" and args.placement == "tuned" and args.placement_candidates > 1:
        try:
            traj, placement = eng.tuned_trajectory_buffers(args.chunk, candidates=args.placement_candidates)
            placement["kind"] = "tuned (timing of candidate sets)"
        except (RuntimeError, MemoryError) as e:   # e.g. out of device memory: measure on the first allocation instead
            torch.cuda.empty_cache()
            traj, placement = eng.trajectory_buffers(args.chunk, layout="separate"), {"error": f"placement tuning failed: {e}"[:300]}
    elif args.placement in ("first", "tuned", "off") or mode != "fused":   # (tuned with < 2 candidates, or a one-launch-per-step mode: nothing to place)
        traj, placement = eng.trajectory_buffers(args.chunk, layout="separate"), {"kind": "first ordinary allocation"}
    else:
        from gym_amd import _native
        big = local_envs * args.chunk * 34 >= _native.SORTED_MIN_BYTES      # 2^17-env shards (8 GPUs) and larger are sorted by HBM class
        try:
            traj = eng.trajectory_buffers(args.chunk, layout=args.placement if big else "separate")
            placement = dict(getattr(eng, "last_placement", None) or {}) if big else {"kind": "ordinary allocations (set below 1 GiB)"}
            if big:
                placement["kind"] = {"sorted": "sorted (ordinary allocations classified with mxv_hbm_pair_probe)",
                                     "placed": "placed (mxv_placed_alloc)"}[args.placement]
        except (RuntimeError, MemoryError) as e:   # e.g. a device someone else is using: measure on ordinary allocations instead of dying
            torch.cuda.empty_cache()
            traj = eng.trajectory_buffers(args.chunk, layout="separate")
            placement = {"kind": "ordinary allocations", "error": f"{args.placement} placement failed: {e}"[:300]}
    trace(f"engine + trajectory tensors ready: {local_envs} envs, placement {placement.get('kind') if placement else None}"
          f" balanced={placement.get('balanced') if placement else None} parked_GiB={placement.get('parked_GiB') if placement else None}")
    launches = [0]
    since_gather = [0]
    issued = [0]
    gathers = [0]
    gathering = world > 1 or args.force_gather

    def run(steps, gather=True):
        """`steps` vector steps as chunk-step launches; at N > 1 the final tensors are all-gathered (asynchronously, overlapping
        the next launch) every time --chunk steps have accumulated — the cadence does not depend on how `steps` was cut."""
        done = 0
        while done < steps:
            k = min(args.chunk, steps - done)
            sr.rollout_per_step(k, mode=mode, out=traj, record_actions=True)
            launches[0] += 1 if mode == "fused" else k
            done += k
            issued[0] += k
            since_gather[0] += k
            if gathering and gather and since_gather[0] >= args.gather_every:
                sr.gather_async()
                since_gather[0] = 0
                gathers[0] += 1

    def fence():
        sr.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Clock state first: the same workload until its rate has settled (time-based, so rank-dependent: no collective inside), THEN
    # reset(seed=0) — which restarts the step index and every env's reset stream — so that everything from here on is a pure
    # function of the arguments and config.work_check is reproducible.
    warm_s, warm_calls = (0.0, 0)
    if args.warm_max_s > 0:
        warm_s, warm_calls = warm_until_stable(lambda: sr.rollout_per_step(args.chunk, mode=mode, out=traj, record_actions=True),
                                               sr.synchronize, max_s=args.warm_max_s)
    trace(f"rate settled after {warm_s:.2f} s ({warm_calls} launches)")
    sr.reset(seed=0)
    # device spin-up: a fixed number of untimed steps; then W warmup steps, which also instantiate the hipGraph(s) and RCCL communicators
    # used in the timed region
    spin = spinup_steps(args.spinup_ms, args.chunk, local_envs)
    run(spin, gather=False)
    fence()   # ranks leave placement and spin-up at different times
    trace(f"spin-up done ({spin} steps), all ranks at the fence")
    since_gather[0] = 0
    run(max(args.warmup, 1))
    if gathering:
        sr.gather()
    fence()
    trace("warm-up done (first gather through the transport included)")

    repeats = args.repeats if args.repeats > 0 else timed_repeats(args.steps, args.chunk, local_envs, args.min_timed_ms, mode)
    timed_steps = args.steps * repeats

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    fence()
    launches[0] = 0
    since_gather[0] = 0
    gathers[0] = 0
    t0 = time.perf_counter()
    ev0.record(eng.stream)
    run(timed_steps)
    ev1.record(eng.stream)
    if gathering:
        sr.wait_gather()
    fence()
    elapsed_local = time.perf_counter() - t0
    trace(f"timed region done: {timed_steps} steps in {elapsed_local * 1e3:.1f} ms")

    launch_ms = ev0.elapsed_time(ev1) / launches[0]  # avg step-kernel launch duration on the engine's stream
    steps_per_launch = timed_steps / launches[0]
    # what the timed region computed, in a form the oracle can reproduce (tests/test_gpu_bench_line.py): the last launch's flags and
    # actions of the first CHECK_ENVS envs, and how many env-steps of that launch ended an episode
    work_check = None
    if rank == 0 and mode == "fused":
        k_last = args.chunk if timed_steps % args.chunk == 0 else timed_steps % args.chunk
        with torch.cuda.stream(eng.stream):
            term, trunc, act = traj["terminated"][:k_last], traj["truncated"][:k_last], traj["actions"][:k_last]
            ended = int(((term | trunc) != 0).sum().item())
            c = min(CHECK_ENVS, local_envs)
            o64 = traj["obs"][:k_last, :c].to(torch.float64)
            work_check = {"what": "last launch of the timed region; checksum = bench.work_checksum(terminated, truncated, actions) over "
                                  f"its {k_last} steps x the first {c} envs; obs_abs_sum / obs_sq_sum / reward_sum over the same block "
                                  "(float64 sums of the float32 observations: the oracle reproduces them to 1e-6 relative, "
                                  "tests/test_gpu_bench_line.py; bit-level observation parity of this very instantiation: tests/test_gpu_soak.py)",
                          "obs_abs_sum": float(o64.abs().sum().item()), "obs_sq_sum": float((o64 * o64).sum().item()),
                          "reward_sum": float(traj["reward"][:k_last, :c].to(torch.float64).sum().item()),
                          "first_step_index": issued[0] - k_last, "steps": k_last, "envs": c,
                          "checksum": work_checksum(term[:, :c], trunc[:, :c], act[:, :c]) if eng.NA > 0 else None,
                          "autoresets_per_env_step": ended / float(k_last * local_envs),
                          "seed": 0, "action_seed": 1}

    # (the probe overwrites the trajectory tensors: it runs after work_check has read them)
    # what THIS rank's placement sustains for the kernel's store pattern with the physics removed (mxv_write_probe, include/mxv.h), into
    # the very tensors the timed region wrote: a rank whose tensors ended up in one HBM class shows here, not only in the job's maximum
    probe_us = None
    if mode == "fused" and not args.compact_outputs and local_envs % 1024 == 0:
        from gym_amd import _native
        torch.cuda.synchronize()
        probe_us = _native.write_probe(local_rank, local_envs, args.chunk, 20, traj["obs"], traj["reward"], traj["actions"],
                                       traj["terminated"], traj["truncated"])
    t = torch.tensor([elapsed_local], dtype=torch.float64, device="cuda")
    per_rank = [{"rank": rank, "device": local_rank, "kernel_us_per_step": launch_ms * 1e3 / steps_per_launch,
                 "timed_region_ms": elapsed_local * 1e3, "write_probe_us_per_step": probe_us,
                 "kernel_over_probe": (launch_ms * 1e3 / steps_per_launch / probe_us) if probe_us else None,
                 "placement": {k: placement.get(k) for k in ("kind", "balanced", "candidates", "parked_GiB", "chunks_created", "class_chunks",
                                                             "seconds", "peak_GiB", "jumped_GiB", "chosen_us_per_step", "error") if k in placement}}]
    if world > 1:
        if args.backend == "gloo":
            tc = t.cpu()
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            t = tc
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    elapsed = float(t.item())

    out = None
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
                "placement": placement,
                "spinup": f"{warm_s:.2f} s of the workload until its rate settled ({warm_calls} launches; before reset(seed=0)), then {spin} "
                          f"untimed steps (nominally {args.spinup_ms:.0f} ms) before the {args.warmup} warmup steps",
                "parallelism": f"env-shard x{world}" + (f", async RCCL all-gather of the final tensors every {args.gather_every} steps "
                                                        f"({args.comm} transport)" if world > 1 else ""),
                "ranks_seen": comm_info.get("ranks_seen", 1),
                "gathers_in_timed_region": gathers[0],
                "gather_every": args.gather_every if gathering else None,
                "gather_transport": (args.comm if gathering else None),
                "comm": comm_info,
                "per_rank": per_rank,
                "work_check": work_check,
                "launch_info": eng.handle.last_launch(),     # mxv_last_launch: the kernel instantiation the timed region ran
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
        if probe_us:
            real_b = 34.0 * local_envs
            out["roofline"]["write_probe"] = {
                "what": "same store pattern, no physics (mxv_write_probe), same tensors",
                "us_per_step": probe_us, "real_GBs": real_b / probe_us / 1e3,
                "kernel_us_per_step": launch_ms * 1e3 / steps_per_launch,
                "kernel_over_probe": launch_ms * 1e3 / steps_per_launch / probe_us}

    sr.close()
    del traj
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_steps)
        if world == 1 and mode == "fused" and not args.compact_outputs and not args.no_variants:
            # reported beside the headline, never instead of it
            v = {}

            def variant(name, fn):       # a failing secondary measurement costs neither the headline nor the other variants
                try:
                    v[name] = fn()
                except Exception as e:  # noqa: BLE001
                    v[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
                    torch.cuda.empty_cache()

            variant("configs2_pendulum", lambda: measure_fused(torch, "Pendulum-v1", 1 << 19, args.chunk, valu=True))
            variant("configs2_mountaincar_continuous", lambda: measure_fused(torch, "MountainCarContinuous-v0", 1 << 19, args.chunk, valu=True))
            variant("mountaincar", lambda: measure_fused(torch, "MountainCar-v0", 1 << 19, args.chunk, valu=True))
            variant("configs3_acrobot_shard", lambda: measure_fused(torch, "Acrobot-v1", 1 << 19, args.chunk, valu=True))
            # the contract-dtype twins (SURVEY.md §8d prices 4-byte rewards and actions; the NumPy adapter widens at the API,
            # gym/vector/sync_vector_env.py:66-71): float32 rewards + int32 actions on the device tensors, every env kind
            variant("compact_cartpole", lambda: measure_fused(torch, ENV_ID, ENVS_TOTAL, args.chunk, compact=True, valu=True))
            variant("compact_pendulum", lambda: measure_fused(torch, "Pendulum-v1", 1 << 19, args.chunk, compact=True))
            variant("compact_mountaincar_continuous", lambda: measure_fused(torch, "MountainCarContinuous-v0", 1 << 19, args.chunk, compact=True))
            variant("compact_mountaincar", lambda: measure_fused(torch, "MountainCar-v0", 1 << 19, args.chunk, compact=True))
            variant("compact_acrobot", lambda: measure_fused(torch, "Acrobot-v1", 1 << 19, args.chunk, compact=True))
            # the wrappers and toy_text engines behind the same library
            variant("normalize", lambda: measure_normalize(torch, ENVS_TOTAL, 128))
            variant("frozenlake8x8", lambda: measure_tabular(torch, "FrozenLake8x8-v1", ENVS_TOTAL, 128))
            variant("taxi", lambda: measure_tabular(torch, "Taxi-v3", ENVS_TOTAL, 128))
            variant("compact_frozenlake8x8", lambda: measure_tabular(torch, "FrozenLake8x8-v1", ENVS_TOTAL, 128, compact=True))
            variant("compact_taxi", lambda: measure_tabular(torch, "Taxi-v3", ENVS_TOTAL, 128, compact=True))
            variant("blackjack", lambda: measure_blackjack(torch, ENVS_TOTAL, 128))
            variant("configs4_mixed_share", lambda: measure_mixed(torch, 1 << 15, args.chunk))
            variant("strong_scaling_share_of_8", lambda: measure_fused(torch, ENV_ID, ENVS_TOTAL // 8, args.chunk))
            variant("step_loop", lambda: {
                "what": "DeviceRollout.step(actions): one launch per vector step with caller-provided actions, 2^20 envs "
                        "(learner-in-the-loop; 66 algorithmic B per env-step; the access pattern without physics: 17.1 us = 0.506, "
                        "profiles/r3j_step_pattern_probe.jsonl)",
                "one_engine": measure_step_loop(torch, ENVS_TOTAL),
                "one_engine_compact": measure_step_loop(torch, ENVS_TOTAL, compact=True),
                "two_half_engines": measure_step_loop(torch, ENVS_TOTAL, halves=2),
                "kernel": measure_step_kernel(torch, ENVS_TOTAL)})
            variant("numpy_loop", lambda: {"num_envs_2^20": measure_numpy_loop(ENVS_TOTAL, 60),
                                           "configs0_num_envs_8": measure_numpy_loop(8, 1000)})    # BASELINE.json configs[0]: the plumbing case
            variant("policy_loop_4096_envs", lambda: measure_policy_loop(torch, 4096))   # last: records a hipGraph
            out["variants"] = v
        print(json.dumps(out), file=json_out, flush=True)

    if world > 1 or solo_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
