#!/usr/bin/env python3
"""vector_make_wrappers_CartPole.npz — `gym.vector.make("CartPole-v1", num_envs=8, asynchronous=False, wrappers=[functools.partial(
TimeLimit, max_episode_steps=7), RecordEpisodeStatistics])` run by THE REFERENCE (gym/vector/__init__.py:56-65: both wrappers around
every sub-env): per step the pre-step fp64 state and the elapsed counter of the OUTER TimeLimit of every sub-env, the actions, the
step's masks and rewards, and what the per-sub-env RecordEpisodeStatistics reported — in infos["final_info"][i]["episode"] (the sub-env is
autoreset in the same step, sync_vector_env.py:152-156): r (float32), l (int32).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_vector_make.py"""
import functools
import os
import sys

import numpy as np

for _name, _val in (("bool8", np.bool_), ("float_", np.float64), ("alltrue", np.all)):
    if not hasattr(np, _name):
        setattr(np, _name, _val)

sys.path.insert(0, "/root/reference")
import gym  # noqa: E402
import warnings  # noqa: E402
from gym.wrappers import RecordEpisodeStatistics, TimeLimit  # noqa: E402

warnings.filterwarnings("ignore")
gym.logger.set_level(gym.logger.ERROR)
HERE = os.path.dirname(os.path.abspath(__file__))


def normalize_case(gid, name, actions_of):
    """wrappers=[TimeLimit(12), NormalizeObservation, NormalizeReward(gamma=0.97), RecordEpisodeStatistics] around every sub-env: the
    per-sub-env running statistics (batches of one; terminal and reset observation of a finished env are two updates), float32 batched
    observations, float64 final observations, normalised rewards, and episode returns that are sums of NORMALISED rewards."""
    from gym.wrappers import NormalizeObservation, NormalizeReward

    N, T, K = 8, 120, 12
    env = gym.vector.make(gid, num_envs=N, asynchronous=False,
                          wrappers=[functools.partial(TimeLimit, max_episode_steps=K), NormalizeObservation,
                                    functools.partial(NormalizeReward, gamma=0.97), RecordEpisodeStatistics])
    obs0, _ = env.reset(seed=321)
    rng = np.random.default_rng(11)
    rec = {k: [] for k in ("state_pre", "elapsed_pre", "action", "obs", "reward", "terminated", "truncated", "final_obs", "ep_r", "ep_l",
                           "raw_obs_post")}

    def raw_obs(e):      # the sub-env's un-normalised observation of its current state (what step / reset returned to the innermost wrapper)
        u = e.unwrapped
        return u._get_obs() if hasattr(u, "_get_obs") else np.array(u.state, dtype=np.float32)

    reset_state = np.array([np.asarray(e.unwrapped.state, dtype=np.float64) for e in env.envs])
    raw_obs0 = np.stack([raw_obs(e) for e in env.envs])
    for t in range(T):
        subs = env.envs
        rec["state_pre"].append(np.array([np.asarray(e.unwrapped.state, dtype=np.float64) for e in subs]))
        tl = [e.env.env.env for e in subs]                  # RecordEpisodeStatistics -> NormalizeReward -> NormalizeObservation -> TimeLimit
        assert all(type(o).__name__ == "TimeLimit" and o._max_episode_steps == K for o in tl)
        rec["elapsed_pre"].append(np.array([o._elapsed_steps for o in tl], dtype=np.int32))
        a = actions_of(rng, N)
        obs, rew, term, trunc, infos = env.step(a)
        assert obs.dtype == np.float32 and rew.dtype == np.float64
        O = obs.shape[1]
        fo, er, el = np.full((N, O), np.nan), np.zeros(N, np.float32), np.zeros(N, np.int32)
        if "final_info" in infos:
            for i, fi in enumerate(infos["final_info"]):
                if fi is not None:
                    assert infos["final_observation"][i].dtype == np.float64
                    fo[i], er[i], el[i] = infos["final_observation"][i], fi["episode"]["r"], fi["episode"]["l"]
        for k, v in (("action", a), ("obs", obs), ("reward", rew), ("terminated", term), ("truncated", trunc), ("final_obs", fo), ("ep_r", er), ("ep_l", el),
                     ("raw_obs_post", np.stack([raw_obs(e) for e in env.envs]))):
            rec[k].append(v)
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(max_episode_steps=np.int64(K), obs0=obs0, reset_state=reset_state, gamma=np.float64(0.97), raw_obs0=raw_obs0)
    # every step's post-reset raw state of the finished envs (the PCG64 reset the reference drew): the replay injects it
    out["state_post"] = np.concatenate([out["state_pre"][1:], np.array([np.asarray(e.unwrapped.state, dtype=np.float64) for e in env.envs])[None]])
    assert (out["truncated"] | out["terminated"]).sum() > 20
    np.savez_compressed(os.path.join(HERE, f"vector_make_normalize_{name}.npz"), **out)
    print(name, "episodes:", int((out["truncated"] | out["terminated"]).sum()))


def clip_action_case():
    """vector_make_clipaction_MountainCarContinuous.npz — `gym.vector.make("MountainCarContinuous-v0", 6, wrappers=ClipAction)` run by THE
    REFERENCE with actions far outside [-1, 1]: the sub-env sees the CLIPPED action (gym/wrappers/clip_action.py:33-43), so the reward's
    penalty `action[0] ** 2 * 0.1` (continuous_mountain_car.py:169) is at most 0.1 — ClipAction is not an identity for this id."""
    from gym.wrappers import ClipAction

    N, T = 6, 60
    env = gym.vector.make("MountainCarContinuous-v0", num_envs=N, asynchronous=False, wrappers=ClipAction)
    env.reset(seed=99)
    rng = np.random.default_rng(3)
    rec = {k: [] for k in ("state_pre", "action", "obs", "reward", "terminated", "truncated")}
    for t in range(T):
        rec["state_pre"].append(np.array([np.asarray(e.unwrapped.state, dtype=np.float64) for e in env.envs]))
        a = (rng.uniform(-6, 6, (N, 1)) * (rng.random((N, 1)) < 0.7) + rng.uniform(-1, 1, (N, 1)) * 0.3).astype(np.float32)
        obs, rew, term, trunc, _ = env.step(a)
        for k, v in (("action", a), ("obs", obs), ("reward", rew), ("terminated", term), ("truncated", trunc)):
            rec[k].append(v)
    out = {k: np.stack(v) for k, v in rec.items()}
    assert (np.abs(out["action"]) > 1).mean() > 0.4 and out["reward"].min() >= -0.1 - 1e-12
    np.savez_compressed(os.path.join(HERE, "vector_make_clipaction_MountainCarContinuous.npz"), **out)
    print("ClipAction: min reward", out["reward"].min(), "out-of-range actions", int((np.abs(out["action"]) > 1).sum()))


def rescale_action_case():
    """vector_make_rescaleaction_{Pendulum,MountainCarContinuous}.npz — `gym.vector.make(id, 6, wrappers=partial(RescaleAction, min_action=a,
    max_action=b))` run by THE REFERENCE with actions inside [a, b]: the sub-env sees low + (high - low) * ((action - a) / (b - a)), clipped
    (gym/wrappers/rescale_action.py:64-83)."""
    from gym.wrappers import RescaleAction

    N, T = 6, 60
    for name, gid, a, b in (("Pendulum", "Pendulum-v1", -1.0, 1.0), ("MountainCarContinuous", "MountainCarContinuous-v0", 0.0, 4.0)):
        env = gym.vector.make(gid, num_envs=N, asynchronous=False, wrappers=functools.partial(RescaleAction, min_action=a, max_action=b))
        assert env.single_action_space.low[0] == a and env.single_action_space.high[0] == b
        env.reset(seed=77)
        rng = np.random.default_rng(5)
        rec = {k: [] for k in ("state_pre", "action", "obs", "reward", "terminated", "truncated")}
        for t in range(T):
            rec["state_pre"].append(np.array([np.asarray(e.unwrapped.state, dtype=np.float64) for e in env.envs]))
            act = rng.uniform(a, b, (N, 1)).astype(np.float32)
            act[rng.random(N) < 0.15] = np.float32(b)      # (the ends of the range: the clip's own cases)
            act[rng.random(N) < 0.15] = np.float32(a)
            obs, rew, term, trunc, _ = env.step(act)
            for k, v in (("action", act), ("obs", obs), ("reward", rew), ("terminated", term), ("truncated", trunc)):
                rec[k].append(v)
        out = {k: np.stack(v) for k, v in rec.items()}
        out.update(min_action=np.float64(a), max_action=np.float64(b))
        np.savez_compressed(os.path.join(HERE, f"vector_make_rescaleaction_{name}.npz"), **out)
        print("RescaleAction", name, "reward range", out["reward"].min(), out["reward"].max())


def transform_case():
    """vector_make_transform_Pendulum.npz — the continuous-control recipe of the PPO scripts people run on gym.vector.make:
    wrappers=[TimeLimit(12), ClipAction, NormalizeObservation, TransformObservation(clip +-1.5), NormalizeReward(gamma=0.97),
    TransformReward(clip +-0.8)] around every sub-env, run by THE REFERENCE (transform_observation.py:34-43, transform_reward.py:36-44 on
    top of normalize.py): the observation transform sees the float64 rows NormalizeObservation returns and the batch rounds to float32
    afterwards; final observations stay float64; the bounds are tight so that both clips act."""
    from gym.wrappers import ClipAction, NormalizeObservation, NormalizeReward, TransformObservation, TransformReward

    N, T, K = 6, 120, 12
    env = gym.vector.make("Pendulum-v1", num_envs=N, asynchronous=False,
                          wrappers=[functools.partial(TimeLimit, max_episode_steps=K), ClipAction, NormalizeObservation,
                                    functools.partial(TransformObservation, f=lambda o: np.clip(o, -1.5, 1.5)),
                                    functools.partial(NormalizeReward, gamma=0.97),
                                    functools.partial(TransformReward, f=lambda r: np.clip(r, -0.8, 0.8))])
    obs0, _ = env.reset(seed=4321)
    rng = np.random.default_rng(13)
    rec = {k: [] for k in ("state_pre", "elapsed_pre", "action", "obs", "reward", "terminated", "truncated", "final_obs", "raw_obs_post")}
    raw_obs = lambda e: e.unwrapped._get_obs()      # noqa: E731
    raw_obs0 = np.stack([raw_obs(e) for e in env.envs])

    def time_limit(e):
        while type(e).__name__ != "TimeLimit" or e._max_episode_steps != K:
            e = e.env
        return e

    for t in range(T):
        rec["state_pre"].append(np.array([np.asarray(e.unwrapped.state, dtype=np.float64) for e in env.envs]))
        rec["elapsed_pre"].append(np.array([time_limit(e)._elapsed_steps for e in env.envs], dtype=np.int32))
        a = rng.uniform(-3, 3, (N, 1)).astype(np.float32)          # (outside [-2, 2] a third of the time: ClipAction's business)
        obs, rew, term, trunc, infos = env.step(a)
        assert obs.dtype == np.float32 and rew.dtype == np.float64
        fo = np.full((N, 3), np.nan)
        if "final_observation" in infos:
            for i, f in enumerate(infos["final_observation"]):
                if f is not None:
                    assert f.dtype == np.float64
                    fo[i] = f
        for k, v in (("action", a), ("obs", obs), ("reward", rew), ("terminated", term), ("truncated", trunc), ("final_obs", fo),
                     ("raw_obs_post", np.stack([raw_obs(e) for e in env.envs]))):
            rec[k].append(v)
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(max_episode_steps=np.int64(K), obs0=obs0, raw_obs0=raw_obs0, gamma=np.float64(0.97), obs_clip=np.float64(1.5), reward_clip=np.float64(0.8))
    assert (np.abs(out["obs"]) == 1.5).mean() > 0.02 and (np.abs(out["reward"]) == 0.8).mean() > 0.02 and out["truncated"].sum() > 20
    np.savez_compressed(os.path.join(HERE, "vector_make_transform_Pendulum.npz"), **out)
    print("transform: clipped obs", float((np.abs(out["obs"]) == 1.5).mean()), "clipped rewards", float((np.abs(out["reward"]) == 0.8).mean()))


def main():
    normalize_case("CartPole-v1", "CartPole", lambda rng, n: (rng.random(n) < np.linspace(0.15, 0.85, n)).astype(np.int64))
    normalize_case("Pendulum-v1", "Pendulum", lambda rng, n: rng.uniform(-2, 2, (n, 1)).astype(np.float32))
    N, T, K = 8, 90, 12
    env = gym.vector.make("CartPole-v1", num_envs=N, asynchronous=False,
                          wrappers=[functools.partial(TimeLimit, max_episode_steps=K), RecordEpisodeStatistics])
    env.reset(seed=123)
    rng = np.random.default_rng(7)
    rec = {k: [] for k in ("state_pre", "elapsed_pre", "action", "obs", "reward", "terminated", "truncated", "ep_mask", "ep_r", "ep_l")}
    for t in range(T):
        subs = env.envs
        rec["state_pre"].append(np.array([np.asarray(e.unwrapped.state, dtype=np.float64) for e in subs]))
        outer = [e.env for e in subs]                       # RecordEpisodeStatistics -> the outer TimeLimit
        assert all(type(o).__name__ == "TimeLimit" and o._max_episode_steps == K for o in outer)
        rec["elapsed_pre"].append(np.array([o._elapsed_steps for o in outer], dtype=np.int32))
        a = (rng.random(N) < np.linspace(0.15, 0.85, N)).astype(np.int64)      # biased pushes: some poles fall before the limit
        obs, rew, term, trunc, infos = env.step(a)
        rec["action"].append(a), rec["obs"].append(obs), rec["reward"].append(rew), rec["terminated"].append(term), rec["truncated"].append(trunc)
        mask, er, el = np.zeros(N, bool), np.zeros(N, np.float32), np.zeros(N, np.int32)
        assert "episode" not in infos                        # nothing at the vector level: the wrapper sits around the sub-envs
        if "final_info" in infos:
            for i, fi in enumerate(infos["final_info"]):
                if fi is not None:
                    assert set(fi) == {"episode"} or set(fi) == {"episode", "TimeLimit.truncated"}, fi
                    ep = fi["episode"]
                    assert isinstance(ep["r"], np.float32) and isinstance(ep["l"], np.int32) and isinstance(ep["t"], float), ep
                    mask[i], er[i], el[i] = True, ep["r"], ep["l"]
        assert np.array_equal(mask, term | trunc)
        rec["ep_mask"].append(mask), rec["ep_r"].append(er), rec["ep_l"].append(el)
    out = {k: np.stack(v) for k, v in rec.items()}
    out["max_episode_steps"] = np.int64(K)
    assert out["truncated"].any() and out["terminated"].any()
    np.savez_compressed(os.path.join(HERE, "vector_make_wrappers_CartPole.npz"), **out)
    print("episodes:", int(out["ep_mask"].sum()), "truncated:", int(out["truncated"].sum()), "terminated:", int(out["terminated"].sum()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "clipaction":      # (added in round 6; the other files are not regenerated)
        clip_action_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "rescaleaction":
        rescale_action_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "transform":
        transform_case()
    else:
        main()
        clip_action_case()
