"""The learner-in-the-loop paths: the NumPy adapter loop, DeviceRollout.step(actions), the loop recorded in a hipGraph, the step kernel alone."""
import time

from .common import *  # noqa: F401,F403
from .common import _event_us, _hbm, _spin


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


def measure_step_loop(torch, envs, steps=600, compact=False, halves=1, obs_carries_state=False):
    """The learner-in-the-loop path: DeviceRollout.step(actions) with caller-provided actions, one launch per vector step
    (gym/vector/sync_vector_env.py:131-169 with a policy in the loop).  halves = 2: the batch as two half-size engines (global env
    indices unchanged: env_offset) on their own streams, stepped alternately — the double-buffered sampling pattern (the policy
    works on one half while the other steps) that lets one half's launch overlap the other's tail."""
    from gym_amd.rollout import DeviceRollout

    n = envs // halves
    eng = [DeviceRollout(ENV_ID, n, env_offset=i * n, seed=0, action_seed=1, reward_f32=compact, action_i32=compact,
                         obs_carries_state=obs_carries_state) for i in range(halves)]
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
    return {"envs": envs, "halves": halves, "obs_carries_state": obs_carries_state, "dtypes": "float32 rewards, int32 actions" if compact else "float64 rewards, int64 actions (the reference's)",
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
