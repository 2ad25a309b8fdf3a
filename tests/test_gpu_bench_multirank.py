"""bench.py's multi-rank control flow (one process per rank, env shards, packed all-gather per chunk, max-over-ranks timing)
run for real with two ranks.  `gpurun` exposes one GPU, so the ranks share it and talk over gloo (`--backend gloo`: same
code path, other transport); what this pins is that every rank issues the same sequence of collectives — a rank-dependent
count (e.g. inside a time-based loop) deadlocks here exactly as it would over RCCL on eight GPUs."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, env=None, timeout=240):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env={**os.environ, **(env or {})})
    assert p.returncode == 0, p.stderr[-2000:]
    return [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_two_ranks_strong_scaling_is_the_default():
    """BASELINE.json's metric: 2^20 logical envs in total, partitioned over the ranks (gather included in the timed region)."""
    lines = _torchrun(["bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "512", "--warmup", "256", "--no-cpu-baseline"])
    assert len(lines) == 1          # rank 0 prints the one line
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 512 and out["scaling"] == "strong"
    cfg = out["config"]
    assert cfg["num_envs_per_gpu"] == 1 << 19 and "num_envs=1048576 (524288 per GPU)" in cfg["workload"]
    assert cfg["timed_steps"] == cfg["repeats"] * 512
    # the final tensors are gathered once per rollout horizon (1024 steps = four launches), not once per launch
    assert cfg["gather_every"] == 1024 and cfg["gathers_in_timed_region"] == cfg["timed_steps"] // 1024 >= 4
    assert out["value"] == pytest.approx((1 << 20) / (out["ms_per_step"] * 1e-3), rel=1e-6)
    assert "cpu_baseline" not in out and out["roofline"]["bound"] == "hbm"
    assert out["roofline"]["env_steps_per_launch"] == pytest.approx((1 << 19) * out["roofline"]["steps_per_launch"])


def test_bench_two_ranks_weak_scaling_and_the_drivers_short_run():
    """--scaling weak keeps 2^20 envs per GPU; `--steps 20 --warmup 5` (what the driver passes) must repeat the 0.1-ms region
    inside one bracket instead of timing a single launch, with the byte accounting of the launches that really ran."""
    lines = _torchrun(["bench.py", "--gpus", "2", "--backend", "gloo", "--scaling", "weak", "--steps", "20", "--warmup", "5",
                       "--no-cpu-baseline"])
    out = lines[0]
    assert out["scaling"] == "weak" and out["config"]["num_envs_per_gpu"] == 1 << 20 and out["steps"] == 20
    cfg, roof = out["config"], out["roofline"]
    assert cfg["repeats"] >= 100 and cfg["timed_steps"] == 20 * cfg["repeats"] and cfg["timed_region_ms"] > 20.0
    assert roof["steps_per_launch"] > 200         # chunk-step launches, not 20-step ones
    b = 4 * 4 + 4 + 4 + 2 + 16.0 * 4 / roof["steps_per_launch"]
    assert roof["algorithmic_bytes_per_env_step"] == pytest.approx(b)
    assert roof["algorithmic_bytes_per_launch"] == pytest.approx(b * (1 << 20) * roof["steps_per_launch"])
    assert roof["traffic"] is None or roof["traffic"] / roof["algorithmic_bytes_per_launch"] < 2.0
    assert out["value"] == pytest.approx((2 << 20) / (out["ms_per_step"] * 1e-3), rel=1e-6)


def test_config_bench_dist_two_ranks_complete():
    lines = _torchrun(["tools/config_bench_dist.py", "--chunk", "64", "--steps", "256"], env={"MXV_DIST_BACKEND": "gloo"})
    assert [l["config"].split(":")[0] for l in lines] == ["config4", "config5"]
    assert all(l["n_gpus"] == 2 for l in lines) and lines[0]["total_envs"] == 1 << 20 and lines[1]["total_envs"] == 1 << 18


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` exactly as the driver calls it for N = 1 — no torchrun: bench.py spawns the two ranks itself
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them) and the ONE line says how many ranks the
    communicator spanned and what every rank measured."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "5"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    # stdout is the ONE line and nothing else: what gloo / RCCL print when a communicator comes up ("[Gloo] Rank 0 is connected ...",
    # "Librccl path : ...") goes to stderr, and so do the ranks' phase markers
    assert len(p.stdout.strip().splitlines()) == 1, p.stdout[:2000]
    assert "[bench rank 1/2" in p.stderr and "timed region done" in p.stderr
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["ranks_seen"] == 2 and cfg["comm"]["launcher"] == "bench.py" and cfg["comm"]["backend"] == "gloo"
    assert cfg["per_rank_fields"] == ["rank", "device", "kernel_us_per_step", "write_probe_us_per_step", "placement", "placement_seconds"]
    assert [r[0] for r in cfg["per_rank"]] == [0, 1] and all(r[2] > 0 and isinstance(r[4], str) for r in cfg["per_rank"])
    # ranks of a multi-process job never take the long placement walk on their own authority, and what they walk is bounded (0.5 s)
    assert all(r[5] is None or r[5] < 1.0 for r in cfg["per_rank"]) and p.stderr.count("placement.seconds") == 2
    assert p.stderr.count("[bench per-rank] ") == 2 and len(p.stdout.encode()) < 4096
    # the older cadence (one gather per launch, rounds 1-3) beside the default one, and the gather by itself
    assert cfg["cadence_ab"]["gather_every"] == 256 and cfg["cadence_ab"]["ms_per_step"] > 0
    assert cfg["gather_us"]["measured_blocking"] > 0 and cfg["gather_us"]["predicted"][0] < cfg["gather_us"]["predicted"][1]
    assert cfg["num_envs_per_gpu"] == 1 << 19 and cfg["repeats"] >= 100
    assert "cpu_baseline" not in out and "variants" not in out          # N = 1 only
