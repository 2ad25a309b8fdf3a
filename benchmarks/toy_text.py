"""the table engine (FrozenLake, Taxi, CliffWalking) and Blackjack."""
import time

from .common import *  # noqa: F401,F403
from .common import _event_us, _hbm, _spin


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


def measure_blackjack(torch, envs, chunk, reps=6, compact=False):
    """Blackjack-v1 (gym/envs/toy_text/blackjack.py:108-160): fused K-step rollouts, observation = three int64 columns, reward float64,
    flags, sampled actions int64 -> 42 B stored per env-step; contract bytes (4-byte scalars): 3 x 4 + 4 + 4 + 2 = 22, which is what
    compact=True (mxv_bj_rollout_compact: int32 / float32) stores."""
    from gym_amd.toy_text import BlackjackRollout

    r = BlackjackRollout(envs, seed=0, action_seed=1, compact=compact)
    r.reset(seed=0)
    out = r.trajectory_buffers(chunk)              # sorted by HBM class, as a caller gets them by default
    run = lambda: r.rollout_per_step(chunk, out=out)   # noqa: E731
    for _ in range(2):
        run()
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    r.synchronize()
    us = (time.perf_counter() - t0) / reps / chunk * 1e6
    stored = 22 if compact else 42
    res = _hbm(us, envs, 22, workload=f"Blackjack-v1, num_envs={envs}, fused {chunk}-step launches, "
                                      + ("int32 observations / actions + float32 rewards" if compact else "the reference's dtypes") + f" ({stored} B stored per env-step)",
               stored_GBs=envs * stored / us / 1e3, episodes_ended_per_env_step=float(((out["terminated"] | out["truncated"]) != 0).float().mean().item()),
               placement=r.last_placement, kernel="bj_kernel (one draw-stream Philox call per step, straight-line)")
    r.close()
    del out
    torch.cuda.empty_cache()
    return res
