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


def measure_toytext_episode_stats(torch, envs, chunk, reps=6):
    """What gym.wrappers.RecordEpisodeStatistics (record_episode_statistics.py:96-151) costs fused into the toy_text trajectory kernels
    (round 6: a float32 add per step + two sparse stores where an episode ended): the same launches with the accumulators off and on,
    both dtype sets.  Sparse output bytes: 8 per finished episode (Blackjack ~ 0.7 per env-step, Taxi ~ 0.005)."""
    from gym_amd.toy_text import BlackjackRollout, TabularRollout

    out = {"workload": f"num_envs={envs}, fused {chunk}-step launches, episode statistics off / on (us per vector step)"}
    for name, make in (("blackjack", lambda c: BlackjackRollout(envs, seed=0, action_seed=1, compact=c)),
                       ("taxi", lambda c: TabularRollout("Taxi-v3", envs, seed=0, action_seed=1, compact=c)),
                       ("frozenlake8x8", lambda c: TabularRollout("FrozenLake8x8-v1", envs, seed=0, action_seed=1, compact=c))):
        for compact in (False, True):
            r = make(compact)
            r.reset(seed=0)
            bufs = r.trajectory_buffers(chunk)
            off = _event_us(torch, r.stream, lambda: r.rollout_per_step(chunk, out=bufs), reps, chunk)
            r.handle.episode_stats(True)
            with torch.cuda.stream(r.stream):
                epr = torch.zeros((chunk, envs), dtype=torch.float32, device=r.device)
                epl = torch.zeros((chunk, envs), dtype=torch.int32, device=r.device)
            r.handle.set_episode_outputs(epr, epl)
            on = _event_us(torch, r.stream, lambda: r.rollout_per_step(chunk, out=bufs), reps, chunk)
            ended = float(((bufs["terminated"] | bufs["truncated"]) != 0).float().mean().item())
            out[f"{name}{'_compact' if compact else ''}"] = {"off_us_per_step": off, "on_us_per_step": on, "ratio": on / off, "episodes_ended_per_env_step": ended}
            r.close()
            del bufs, epr, epl
            torch.cuda.empty_cache()
    k = out["blackjack"]
    out["us_per_step"] = k["on_us_per_step"]
    return out
