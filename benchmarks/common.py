"""Shared pieces of the bench: constants of the roofline, the algorithmic-bytes contract, the pure functions that fix the
launch shape of the timed region, timing helpers, and the readers of the committed PMC passes under profiles/."""
import json
import math
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def _spin(fn, sync, ms):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        fn()
        sync()


def warm_until_stable(fn, sync, max_s=4.0, window_s=0.1, tol=0.01, min_s=0.6):
    """Run fn until its rate has settled: successive windows of window_s agree within tol (and at least min_s have passed), or max_s.
    A box that has been idle needs more than a second of load before its clocks, and with them the kernel's instruction stream, reach
    their sustained state (the first process on a fresh box measured 6.4-6.5 us per step after 0.2 s of load, 5.8 after 3 s;
    profiles/r3/r3d_*).  Returns (seconds, calls)."""
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
