"""Fused trajectory launches of one env kind (BASELINE.json configs[1..3]) and the mixed batch of configs[4], one GPU."""
import time

from .common import *  # noqa: F401,F403
from .common import _event_us, _hbm, _spin


def measure_fused(torch, env_id, envs, chunk, *, compact=False, launches=8, spin_ms=60.0, probe=True, valu=False, placement_mode=None):
    """One env kind, fused trajectory launches on one GPU: us per step, env-steps/s, roofline on the algorithmic bytes, the
    write probe of its own store pattern into the same tensors.  placement_mode: MXV_PLACEMENT for the allocation of the trajectory
    tensors ("cheap": at most 8 GiB parked; "search": the long walk; default: the process's setting — auto: the long walk only on an
    otherwise empty device)."""
    import os

    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout(env_id, envs, seed=0, action_seed=1, reward_f32=compact, action_i32=compact)
    r.reset(seed=0)
    saved = os.environ.get("MXV_PLACEMENT")
    if placement_mode is not None and saved not in ("off", "0", "no", "false"):
        os.environ["MXV_PLACEMENT"] = placement_mode
    try:
        traj = r.trajectory_buffers(chunk)
    finally:
        if saved is None:
            os.environ.pop("MXV_PLACEMENT", None)
        else:
            os.environ["MXV_PLACEMENT"] = saved
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
