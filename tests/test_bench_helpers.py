"""bench.py's host logic that needs no GPU: the launcher fails fast and readably, the step counts of the untimed phases are pure
functions of the arguments (config.work_check depends on it), the checksum has one definition on both sides (torch / NumPy), and the
reference's CPU baseline comes from a committed run of tools/reference_baseline.py, not from a literal."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

from conftest import HAS_GPU, ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_work_checksum_is_the_same_function_on_torch_and_numpy():
    rng = np.random.default_rng(3)
    K, n = 7, 129
    term = rng.integers(0, 2, (K, n)).astype(np.uint8)
    trunc = rng.integers(0, 2, (K, n)).astype(np.uint8)
    act = rng.integers(0, 3, (K, n)).astype(np.int64)
    a = bench.work_checksum(term, trunc, act)
    b = bench.work_checksum(torch.from_numpy(term), torch.from_numpy(trunc), torch.from_numpy(act))
    ref = 0
    for k in range(K):
        for i in range(n):
            ref = (ref + (int(term[k, i]) + 2 * int(trunc[k, i]) + 4 * int(act[k, i])) * (((k * n + i) * 0x9E3779B97F4A7C15 + 1) % 2**64)) % 2**64
    assert a == b == ref
    act[3, 5] ^= 1
    assert bench.work_checksum(term, trunc, act) != a


def test_untimed_phases_are_pure_functions_of_the_arguments():
    assert bench.spinup_steps(150.0, 256, 1 << 20) == bench.spinup_steps(150.0, 256, 1 << 20) == 98 * 256
    assert bench.spinup_steps(0.0, 256, 1 << 20) == 0
    assert bench.spinup_steps(150.0, 256, 1 << 17) % 256 == 0
    r = bench.timed_repeats(20, 256, 1 << 20, 60.0)
    assert r == 512 and (r * 20) % 256 == 0          # the driver's --steps 20: 40 launches of 256 steps
    assert bench.timed_repeats(20480, 256, 1 << 20, 60.0) == 1
    assert bench.algorithmic_bytes_per_env_step("fused", 256) == 26.25 and bench.algorithmic_bytes_per_env_step("given", 1) == 66
    assert bench.algorithmic_bytes_per_env_step("fused", 256, "Acrobot-v1") == 34.25


def test_reference_baseline_is_a_committed_measurement_not_a_literal():
    j = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu_baseline.json")))
    assert j["reference_version"] == "0.26.2" and j["host"]["numpy"]
    assert set(j["single_core"]) == {"CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0", "MountainCarContinuous-v0"}
    assert set(j["single_core"]["CartPole-v1"]) == {"8", "64", "1024"}
    assert 2e4 < j["headline"]["value"] < 1e6 and j["all_cores"]["processes"] >= 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "8.0e4" not in src and "reference_cpu_baseline.json" in src
    ref = bench.reference_python_baseline()
    assert ref["value"] == j["headline"]["value"] and ref["kind"] == "reference"
    if os.path.isdir("/root/reference/gym"):      # the build container: the reference is re-timed live beside the committed figure
        assert ref["live"]["value"] == __import__("pytest").approx(ref["value"], rel=0.6)


def test_self_launch_fails_fast_and_readably_when_a_rank_cannot_start():
    """`python bench.py --gpus 2` with no launcher starts its own ranks; here (no HIP device, or a bad argument on a GPU box) every
    rank exits at once: the parent must return non-zero within seconds with the reason on stderr — not hang in a rendezvous."""
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "5",
            "--no-cpu-baseline", "--launch-timeout", "120"]
    if HAS_GPU:
        args += ["--chunk", "0"]       # makes every rank fail after start-up
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    p = subprocess.run(args, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and time.time() - t0 < 120
    assert "rank" in p.stderr and ("HIP device" in p.stderr or "exited with code" in p.stderr)
    assert not any(l.startswith("{") for l in p.stdout.splitlines())
