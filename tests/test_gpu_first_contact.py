"""First contact with an 8-GPU node, rehearsed on the one GPU gpurun exposes (Next #5): no scaling curve can be measured
here, so everything that can go wrong BEFORE the first collective carries data is made boring in advance.

  * `python bench.py --gpus 8` exactly as the driver calls it, except that the eight self-launched ranks share the one device and talk
    over gloo: the launcher, the rendezvous, the rank-symmetric sequence of collectives (a rank-dependent count deadlocks here as it
    would over RCCL), the per-rank report (kernel time, write probe of the rank's own placement) and the shard shape (2^17 envs: one env
    per lane) are the real ones;
  * the gather's transport itself — the C ABI's RCCL communicator, ncclAllGather per output tensor on the side stream — forced at world
    size 1 inside bench.py's timed region (--force-gather);
  * placement beside a process that holds 200 GiB of the device: trajectory_buffers() must not raise, and MXV_PLACEMENT=off / bench.py
    --placement off must not even probe."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "MXV_PLACEMENT", "MXV_PLACEMENT_MAX_PARK_GIB")


def _env(**extra):
    return dict({k: v for k, v in os.environ.items() if k not in CLEAN}, **extra)


def _one_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(p.stdout.strip().splitlines()) == 1, p.stdout[:2000]
    return lines[0]


def test_bench_with_eight_self_launched_ranks():
    # (round 6: ONE eight-rank rehearsal, sized to stay under 30 s — a chunk of 64 steps, one repeat, one gather in the timed region; BASELINE.json
    # no cadence A/B over gloo beyond two ranks; configs[3] / configs[4] run their rank code with two ranks in tests/test_gpu_bench_multirank.py and their eight-rank partition
    # arithmetic in tests/test_gpu_configs.py: eight more processes on one GPU added 21 s and no coverage)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--backend", "gloo", "--chunk", "64", "--steps", "128", "--warmup", "64", "--repeats", "1",
                        "--gather-every", "128", "--placement", "off", "--warm-max-s", "0.2", "--spinup-ms", "10"], cwd=ROOT, capture_output=True, text=True,
                       timeout=300, env=_env())
    out = _one_line(p)
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["ranks_seen"] == 8 and cfg["comm"]["launcher"] == "bench.py" and out["scaling"] == "strong"
    assert cfg["num_envs_per_gpu"] == 1 << 17 and "num_envs=1048576 (131072 per GPU)" in cfg["workload"]
    assert len(p.stdout.encode()) < 4096, len(p.stdout)           # the 8-rank line fits the driver's record as well
    assert cfg["per_rank_fields"] == ["rank", "device", "kernel_us_per_step", "write_probe_us_per_step", "placement", "placement_seconds"]
    assert [r[0] for r in cfg["per_rank"]] == list(range(8))
    for r in cfg["per_rank"]:                                   # every rank says what it measured on ITS tensors
        assert r[2] > 0 and r[3] > 0, r     # (eight ranks share the device here: the ratio means nothing, its presence does)
    assert p.stderr.count("[bench per-rank] ") == 8
    assert cfg["gathers_in_timed_region"] == 1 and cfg["gather_transport"] == "torch"      # (a gloo gather of 8 ranks sharing one GPU takes seconds)
    # the gather by itself beside the link model: 7 x 3.4 MB received per rank at 80-150 GB/s
    g = cfg["gather_us"]
    assert g["bytes_received_per_rank"] >= 7 * (1 << 17) * 26 and g["measured_blocking"] > 0 and 100 < g["predicted"][0] < g["predicted"][1] < 500, g
    li = cfg["launch_info"]                                     # the strong-scaling shard runs the one-env-per-lane instantiation
    assert (li["kernel"], li["envs_per_lane"], li["safe"], li["out_mode"], li["grid"]) == (1, 1, 0, 1, (1 << 17) // 64), li
    assert cfg["placement"]["kind"] == "first ordinary allocation"
    assert out["value"] == pytest.approx((1 << 20) / (out["ms_per_step"] * 1e-3), rel=1e-6)
    for rank in range(8):
        assert f"[bench rank {rank}/8" in p.stderr
    assert "cpu_baseline" not in out and "variants" not in out


@pytest.mark.parametrize("comm", ["mxv", "torch"])
def test_the_gather_inside_the_timed_region_at_world_size_1(comm):
    """--comm mxv: libmxv.so opens RCCL itself, builds a one-rank communicator and issues the grouped ncclAllGather calls of every chunk
    on its side stream, overlapping the next launch — the code an 8-GPU run executes, minus the links."""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--comm", comm, "--force-gather", "--gather-every", "256", "--steps", "1024", "--warmup", "256",
                        "--no-cpu-baseline", "--no-variants", "--placement", "off", "--warm-max-s", "0.4"], cwd=ROOT, capture_output=True,
                       text=True, timeout=300, env=_env())
    out = _one_line(p)
    cfg = out["config"]
    assert cfg["gather_transport"] == comm and cfg["gathers_in_timed_region"] == cfg["timed_steps"] // 256 >= 4
    if comm == "torch":      # a real one-rank NCCL (= RCCL) process group: init_process_group, the first all-reduce, all_gather_into_tensor
        assert cfg["comm"]["backend"] == "nccl" and cfg["comm"]["ranks_seen"] == 1 and cfg["comm"]["rccl_version"]
    assert out["n_gpus"] == 1 and out["value"] > 1e11        # the gather's launch path costs percents, not factors
    assert out["roofline"]["frac"] > 0.4


def test_placement_beside_a_process_that_holds_most_of_the_device():
    """200 GiB of the 288 belong to THIS process; a second one asks for the 9-GiB trajectory set of the headline configuration with the
    default layout (sorted by HBM class: the search parks memory while it probes).  It must come back with tensors — balanced or not —
    and with less parked than half of what was left; with MXV_PLACEMENT=off it must not probe at all."""
    import torch

    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info(0)
    hold = min(200 << 30, free - (48 << 30))
    assert hold > (64 << 30), f"only {free >> 30} GiB free on the device"
    hog = torch.empty(hold, dtype=torch.uint8, device="cuda:0")
    hog[:: 1 << 20].fill_(1)
    torch.cuda.synchronize()
    code = ("import json, torch\n"
            "from gym_amd.rollout import DeviceRollout\n"
            "free0 = torch.cuda.mem_get_info(0)[0]\n"
            "r = DeviceRollout('CartPole-v1', 1 << 20, seed=0, action_seed=1); r.reset(seed=0)\n"
            "t = r.trajectory_buffers(256)\n"
            "out = r.rollout_per_step(256, out=t); r.synchronize()\n"
            "print(json.dumps({'free0_GiB': free0 / 2**30, 'placement': getattr(r, 'last_placement', None),\n"
            "                  'bytes': sum(x.numel() * x.element_size() for x in t.values()), 'ended': int((out['terminated'] | out['truncated']).sum())}))\n")
    try:
        for extra, probing in (({}, True), ({"MXV_PLACEMENT": "off"}, False), ({"MXV_PLACEMENT_MAX_PARK_GIB": "2"}, True)):
            p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300, env=_env(**extra))
            assert p.returncode == 0, (extra, p.stderr[-3000:])
            j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            assert j["bytes"] == (1 << 20) * 256 * 34 and j["ended"] > 0
            assert j["free0_GiB"] < (total - hold) / 2**30 + 1
            if not probing:
                assert j["placement"] is None
                continue
            rep = j["placement"]
            assert rep["kind"] == "sorted" and rep["parked_GiB"] <= rep.get("budget_GiB", 0) + 0.01, rep
            assert rep.get("budget_GiB", 0) <= (j["free0_GiB"] - 8.5) / 2 + 0.5
            if not extra:       # the default beside somebody who holds most of the device: the 8-GiB cap, not the long walk
                assert rep["mode"] == "auto->cheap" and rep["parked_GiB"] <= 8.0 and rep["budget_GiB"] <= 8.0, rep
            if "MXV_PLACEMENT_MAX_PARK_GIB" in extra:
                assert rep["budget_GiB"] <= 2.0 and rep["parked_GiB"] <= 2.0
    finally:
        del hog
        torch.cuda.empty_cache()
