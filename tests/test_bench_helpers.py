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


def _full_record():
    """A long-form record as bench.py builds it: round 4's committed driver-argument run (24.9 KB as ONE line then: the driver's 8-KB
    record lost the headline), plus the fields round 5 added."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r4", "r4w_bench_driver_args_final_tree.json")))
    full["cpu_baseline"]["sample_short"] = "C port (gcc -O2): 1 thread 2^20 envs x 100 steps 3.1 s; 64 threads x 16384 envs x 3000 steps 2.2 s"
    full["details"] = {"variants": "gpurun_out/bench_variants.json", "headline": "gpurun_out/bench_headline.json"}
    full["config"].update(gather_us=None, cadence_ab=None)
    return full


def test_compact_line_fits_the_drivers_record_and_keeps_the_contract():
    full = _full_record()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    s = json.dumps(line)
    assert len(s) < bench.LINE_LIMIT == 4096 and json.loads(s) == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k] or line[k] == __import__("pytest").approx(full[k], rel=1e-7), k
    assert line["value"] == full["value"]                                    # the headline figure is never rounded
    roof = line["roofline"]
    assert (roof["bound"], roof["peak"], roof["unit"]) == ("hbm", 8000.0, "GB/s") and roof["frac"] == __import__("pytest").approx(full["roofline"]["frac"], rel=1e-7)
    assert roof["env_steps_per_launch"] == 1 << 28 and roof["kernel_over_probe"] > 1.0       # whole numbers stay exact
    assert line["config"]["work_check"]["checksum"] == full["config"]["work_check"]["checksum"] and "what" not in line["config"]["work_check"]
    assert line["config"]["launch_info"] == full["config"]["launch_info"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["reference_python"]["value"] > 1e4
    assert set(line["variants"]) == set(full["variants"]) and all(v == "error" or len(v) == 2 for v in line["variants"].values())
    assert line["variants"]["step_loop"][1] == __import__("pytest").approx(full["variants"]["step_loop"]["one_engine"]["roofline"]["frac"], rel=1e-3)


def test_compact_line_of_an_eight_rank_run():
    import copy

    f8 = copy.deepcopy(_full_record())
    f8.pop("variants"), f8.pop("cpu_baseline")
    f8["n_gpus"] = 8
    r0 = f8["config"]["per_rank"][0]
    f8["config"]["per_rank"] = [dict(r0, rank=i, device=i, kernel_us_per_step=0.78 + 0.0123456789 * i) for i in range(8)]
    f8["config"]["gather_us"] = {"measured_blocking": 312.123456789, "predicted": [180.123456, 320.987654], "bytes_received_per_rank": 24117248,
                                 "steps_of_this_shard_it_equals": 400.123456}
    f8["config"]["cadence_ab"] = {"gather_every": 256, "steps": 2560, "ms_per_step": 0.00112345678}
    f8["config"]["comm"] = {"backend": "nccl", "ranks_seen": 8, "transport": "torch", "launcher": "bench.py", "rccl_version": "2.26.6"}
    line = bench.compact_line(f8)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert [r[0] for r in line["config"]["per_rank"]] == list(range(8)) and line["config"]["per_rank"][3][4] == "sorted"
    assert line["config"]["gather_us"]["bytes_received_per_rank"] == 24117248 and line["config"]["cadence_ab"]["gather_every"] == 256
    # a pathological record (a placement error message on every rank ...) sheds its optional parts instead of outgrowing the limit
    f8["config"]["outputs"] = "x" * 3000
    assert len(json.dumps(bench.compact_line(f8))) < bench.LINE_LIMIT
