#!/usr/bin/env python3
"""tools/reference_baseline.py — time the REFERENCE itself (SURVEY.md §8d "CPU baseline beside it").

What is timed: /root/reference's gym.vector.SyncVectorEnv (gym/vector/sync_vector_env.py:135-169), imported under the NumPy-2
alias shim of SURVEY App. C, `gym.vector.make(id, num_envs=n, asynchronous=False, disable_env_checker=True)`, `reset(seed=0)`,
`action_space.seed(0)`, 50 warm-up steps, then 1000 x `step(action_space.sample())` under time.perf_counter, best of 3;
n in {8, 64, 1024}; the five classic-control ids.  SyncVectorEnv is a serial Python loop -> one core.  The all-core aggregate is
one independent SyncVectorEnv process per available core (CartPole-v1, n = 64), all timing the same 1000 steps behind a barrier.

    python tools/reference_baseline.py                   # -> profiles/reference_cpu_baseline.json
    python tools/reference_baseline.py --quick --stdout  # CartPole-v1 only, n = 64, best of 1 (what bench.py re-times live)

bench.py reads profiles/reference_cpu_baseline.json for `cpu_baseline.reference_python` (no literal), and calls
`measure(quick=True)` itself when the reference tree is importable where it runs (the build container; not the GPU box).
Nothing under gym_amd/ imports this file.
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("GYM_REFERENCE_PATH", "/root/reference")
IDS = ["CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0"]


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE, "gym", "vector", "sync_vector_env.py"))


def _import_reference():
    """the aliases NumPy 2 removed, then the reference from its read-only tree."""
    import numpy as np

    for name, val in (("bool8", np.bool_), ("float_", np.float64), ("alltrue", np.all)):
        if not hasattr(np, name):
            setattr(np, name, val)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import warnings

    warnings.filterwarnings("ignore")
    import gym

    assert os.path.abspath(gym.__file__).startswith(os.path.abspath(REFERENCE)), gym.__file__
    return gym


def time_sync_vector_env(env_id: str, n: int, steps: int = 1000, warmup: int = 50, best_of: int = 3, barrier=None) -> float:
    """env-steps/s of gym.vector.SyncVectorEnv(env_id) x n, one process, random actions from action_space.sample()."""
    gym = _import_reference()
    env = gym.vector.make(env_id, num_envs=n, asynchronous=False, disable_env_checker=True)
    assert type(env).__name__ == "SyncVectorEnv"
    env.reset(seed=0)
    env.action_space.seed(0)
    for _ in range(warmup):
        env.step(env.action_space.sample())
    best = float("inf")
    for _ in range(best_of):
        if barrier is not None:
            barrier.wait()
        t0 = time.perf_counter()
        for _ in range(steps):
            env.step(env.action_space.sample())
        best = min(best, time.perf_counter() - t0)
    env.close()
    return n * steps / best


def _proc(env_id, n, steps, best_of, barrier, q):
    q.put(time_sync_vector_env(env_id, n, steps=steps, best_of=best_of, barrier=barrier))


def all_core_aggregate(env_id: str = "CartPole-v1", n: int = 64, steps: int = 1000, best_of: int = 3, procs: int = 0):
    """One independent SyncVectorEnv process per available core, each timing the same steps behind a barrier;
    aggregate = sum of the per-process rates (every process runs its own best-of)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    procs = procs or avail
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=_proc, args=(env_id, n, steps, best_of, barrier, q)) for _ in range(procs)]
    for p in ps:
        p.start()
    rates = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(60)
    return {"processes": procs, "cores_available": avail, "env_id": env_id, "num_envs_per_process": n,
            "value": sum(rates), "per_process_min": min(rates), "per_process_max": max(rates), "unit": "env-steps/s"}


def host_description() -> dict:
    import numpy as np

    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return {"cpu": cpu, "cores_available": avail, "os_cpu_count": os.cpu_count(), "python": platform.python_version(),
            "numpy": np.__version__, "machine": platform.machine()}


def measure(quick: bool = False) -> dict:
    if not reference_available():
        raise FileNotFoundError(f"{REFERENCE}/gym is not present here")
    gym = _import_reference()
    out = {
        "what": "gym.vector.SyncVectorEnv of the reference (gym/vector/sync_vector_env.py:135-169) under the NumPy-2 alias shim, "
                "reset(seed=0), action_space.seed(0), 50 warm-up + 1000 timed step(action_space.sample()), time.perf_counter, "
                + ("best of 1 (quick)" if quick else "best of 3") + "; one process = one core",
        "reference_version": gym.__version__,
        "host": host_description(),
        "unit": "env-steps/s",
        "single_core": {},
    }
    ids, sizes, best_of = (["CartPole-v1"], [64], 1) if quick else (IDS, [8, 64, 1024], 3)
    for env_id in ids:
        out["single_core"][env_id] = {str(n): time_sync_vector_env(env_id, n, best_of=best_of) for n in sizes}
    out["headline"] = {"env_id": "CartPole-v1", "num_envs": 64, "value": out["single_core"]["CartPole-v1"]["64"],
                       "unit": "env-steps/s/core"}
    if not quick:
        out["all_cores"] = all_core_aggregate()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--stdout", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_cpu_baseline.json"))
    args = ap.parse_args()
    res = measure(quick=args.quick)
    res["measured_at"] = time.strftime("%Y-%m-%d %H:%M:%S")
    text = json.dumps(res, indent=1)
    if args.stdout:
        print(text)
    else:
        with open(args.out, "w") as f:
            f.write(text + "\n")
        print(f"wrote {args.out}: CartPole-v1 n=64 {res['headline']['value']:.3e} env-steps/s/core"
              + (f", all cores ({res['all_cores']['processes']} processes) {res['all_cores']['value']:.3e}" if "all_cores" in res else ""))


if __name__ == "__main__":
    main()
