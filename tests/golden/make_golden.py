#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference (openai/gym 0.26.2) is imported read-only from /root/reference under this
container's NumPy 2.2.6 + glibc 2.35 with the NumPy-2 alias shim of SURVEY.md App. C.
Two kinds of vectors per env id (SURVEY.md §8c parity protocol):

  P1  single raw-env steps (`env.unwrapped.step`) from hand-set fp64 states sampled broadly and
      near every threshold, with in- and out-of-range actions.            -> <env>_p1.npz
  P2  trajectories of `gym.vector.make(id, num_envs=8, asynchronous=False)` (SyncVectorEnv over
      TimeLimit(OrderEnforcing(PassiveEnvChecker(env)))) with seeded random actions: the pre-step
      fp64 state/elapsed of every sub-env, the actions, every output incl. final_observation,
      and the post-step state/elapsed (after the reference's own PCG64 autoreset). -> <env>_p2.npz

Arrays are small (a few hundred KB total).  The script is deterministic (fixed seeds).
"""
import os
import sys

import numpy as np

for _name, _val in (("bool8", np.bool_), ("float_", np.float64), ("alltrue", np.all)):
    if not hasattr(np, _name):
        setattr(np, _name, _val)
if not hasattr(np, "cast"):
    class _Cast:
        def __getitem__(self, dt):
            return lambda v: np.asarray(v, dtype=dt)
    np.cast = _Cast()

sys.path.insert(0, "/root/reference")
import gym  # noqa: E402
import warnings  # noqa: E402

warnings.filterwarnings("ignore")
gym.logger.set_level(gym.logger.ERROR)

HERE = os.path.dirname(os.path.abspath(__file__))

ENVS = {
    # name: (gym id, state dim, obs dim, discrete n or 0, short horizon for a truncation-heavy P2 run)
    "CartPole": ("CartPole-v1", 4, 4, 2, 9),
    "Pendulum": ("Pendulum-v1", 2, 3, 0, 7),
    "Acrobot": ("Acrobot-v1", 4, 6, 3, 60),
    "MountainCar": ("MountainCar-v0", 2, 2, 3, 50),
    "MountainCarContinuous": ("MountainCarContinuous-v0", 2, 2, 0, 80),
}


def set_state(raw, name, s, fresh):
    """Write fp64 state `s` into the raw reference env with the container type the env itself uses."""
    if name == "CartPole":
        raw.state = tuple(float(v) for v in s)  # cartpole.py:160 keeps a tuple of floats
        raw.steps_beyond_terminated = None
    elif name == "MountainCarContinuous":
        # float64 array right after reset (:182), float32 array after any step (:171)
        raw.state = np.array(s, dtype=np.float64 if fresh else np.float32)
    else:
        raw.state = np.array(s, dtype=np.float64)


def get_state(raw):
    return np.asarray(raw.state, dtype=np.float64).copy()


def p1_states(name, rng, n):
    """Broad + near-threshold fp64 states."""
    if name == "CartPole":
        s = np.stack([rng.uniform(-2.6, 2.6, n), rng.uniform(-3, 3, n),
                      rng.uniform(-0.25, 0.25, n), rng.uniform(-3.5, 3.5, n)], 1)
        k = n // 4
        s[:k, 0] = np.sign(rng.uniform(-1, 1, k)) * (2.4 + rng.uniform(-0.07, 0.07, k))
        s[k:2 * k, 2] = np.sign(rng.uniform(-1, 1, k)) * (0.20943951023931953 + rng.uniform(-0.07, 0.07, k))
        s[2 * k:2 * k + 64] = rng.uniform(-0.05, 0.05, (64, 4))
    elif name == "Pendulum":
        s = np.stack([rng.uniform(-90, 90, n), rng.uniform(-8, 8, n)], 1)
        k = n // 4
        s[:k, 0] = rng.uniform(-np.pi, np.pi, k)
        s[k:2 * k, 1] = np.sign(rng.uniform(-1, 1, k)) * (8 - rng.uniform(0, 0.3, k))
        s[2 * k:2 * k + 8, 0] = [0.0, np.pi, -np.pi, 2 * np.pi, -3 * np.pi, 1e-9, -1e-9, 5 * np.pi]
    elif name == "Acrobot":
        s = np.stack([rng.uniform(-np.pi, np.pi, n), rng.uniform(-np.pi, np.pi, n),
                      rng.uniform(-4 * np.pi, 4 * np.pi, n), rng.uniform(-9 * np.pi, 9 * np.pi, n)], 1)
        k = n // 4
        s[:k] = rng.uniform(-0.1, 0.1, (k, 4)).astype(np.float32)  # reset-like (float32-rounded)
        s[k:2 * k, 0] = np.pi - rng.uniform(0, 0.6, k)             # near the height threshold
        s[k:2 * k, 1] = rng.uniform(-0.5, 0.5, k)
        s[2 * k:3 * k, 2] = np.sign(rng.uniform(-1, 1, k)) * (4 * np.pi - rng.uniform(0, 0.5, k))
        s[2 * k:3 * k, 3] = np.sign(rng.uniform(-1, 1, k)) * (9 * np.pi - rng.uniform(0, 0.5, k))
    else:  # MountainCar / MountainCarContinuous
        s = np.stack([rng.uniform(-1.2, 0.6, n), rng.uniform(-0.07, 0.07, n)], 1)
        k = n // 5
        goal = 0.5 if name == "MountainCar" else 0.45
        s[:k, 0] = -1.2 + rng.uniform(0, 0.08, k)
        s[:k, 1] = -rng.uniform(0, 0.07, k)
        s[k:2 * k, 0] = goal + rng.uniform(-0.08, 0.08, k)
        s[2 * k:3 * k, 0] = 0.6 - rng.uniform(0, 0.08, k)
        s[2 * k:3 * k, 1] = rng.uniform(0, 0.07, k)
        s[3 * k:4 * k, 1] = np.sign(rng.uniform(-1, 1, k)) * (0.07 - rng.uniform(0, 0.002, k))
        s[4 * k:4 * k + 3] = [[-1.2, -0.01], [-1.2, 0.0], [goal, 0.0]]
    return s


def p1_actions(name, rng, n):
    gid, S, O, nd, _ = ENVS[name]
    if nd:
        return rng.integers(0, nd, n).astype(np.int64)
    lim = 2.0 if name == "Pendulum" else 1.0
    a = rng.uniform(-lim, lim, n).astype(np.float32)
    k = n // 4
    a[:k] = rng.uniform(-3 * lim, 3 * lim, k).astype(np.float32)  # out of bounds -> clip path
    a[k:k + 4] = np.array([lim, -lim, 0.0, np.nextafter(np.float32(lim), np.float32(9))], dtype=np.float32)
    return a


def make_p1(name, n=4096, seed=20260921, save=True):
    """save=False: return the vectors instead of writing the fixture (tests/test_oracle_live_reference.py: fresh seeds against the live
    reference, where it exists)."""
    gid, S, O, nd, _ = ENVS[name]
    rng = np.random.default_rng(seed + sum(map(ord, name)))
    raw = gym.make(gid, disable_env_checker=True).unwrapped
    raw.reset(seed=0)
    s0 = p1_states(name, rng, n)
    act = p1_actions(name, rng, n)
    fresh = np.zeros(n, dtype=np.uint8)
    if name == "MountainCarContinuous":
        fresh[: n // 2] = 1
        s0[n // 2:] = s0[n // 2:].astype(np.float32)  # a float32 state is what the env holds when not fresh
    obs = np.zeros((n, O), np.float32)
    rew = np.zeros(n, np.float64)
    term = np.zeros(n, np.uint8)
    s1 = np.zeros((n, S), np.float64)
    for i in range(n):
        set_state(raw, name, s0[i], bool(fresh[i]))
        a = act[i] if nd else np.array([act[i]], dtype=np.float32)
        o, r, te, tr, info = raw.step(a)
        assert tr is False and info == {}
        obs[i] = o
        rew[i] = r
        term[i] = te
        s1[i] = get_state(raw)
    if not save:
        return dict(state0=s0, action=act, fresh=fresh, obs=obs, reward=rew, terminated=term, state1=s1)
    np.savez_compressed(os.path.join(HERE, f"{name}_p1.npz"), state0=s0, action=act, fresh=fresh, obs=obs,
                        reward=rew, terminated=term, state1=s1)
    print(f"{name:24s} P1: {n} steps, {int(term.sum())} terminations")


def make_p2(name, tag, T, max_episode_steps=None, num_envs=8, seed=123, save=True):
    gid, S, O, nd, _ = ENVS[name]
    kwargs = {} if max_episode_steps is None else {"max_episode_steps": max_episode_steps}
    venv = gym.vector.make(gid, num_envs=num_envs, asynchronous=False, **kwargs)
    limit = venv.envs[0]._max_episode_steps
    venv.action_space.seed(seed + 1)
    # the reference's RecordEpisodeStatistics rides along (it does not alter the stream)
    stats = gym.wrappers.RecordEpisodeStatistics(venv)
    obs0, _ = stats.reset(seed=seed)
    N = num_envs
    rec = dict(
        state_pre=np.zeros((T, N, S)), elapsed_pre=np.zeros((T, N), np.int32),
        action=np.zeros((T, N), np.int64 if nd else np.float32),
        obs=np.zeros((T, N, O), np.float32), reward=np.zeros((T, N)), terminated=np.zeros((T, N), np.uint8),
        truncated=np.zeros((T, N), np.uint8), final_obs=np.zeros((T, N, O), np.float32),
        final_mask=np.zeros((T, N), np.uint8), state_post=np.zeros((T, N, S)),
        elapsed_post=np.zeros((T, N), np.int32),
        ep_return=np.zeros((T, N), np.float32), ep_length=np.zeros((T, N), np.int32), ep_mask=np.zeros((T, N), np.uint8),
        ep_running_return=np.zeros((T, N), np.float32),
    )
    for t in range(T):
        for i, e in enumerate(venv.envs):
            rec["state_pre"][t, i] = get_state(e.unwrapped)
            rec["elapsed_pre"][t, i] = e._elapsed_steps
        a = venv.action_space.sample()
        if not nd and t % 7 == 3:  # exercise the clip path inside real trajectories too
            a = (a * 2.5).astype(np.float32)
        o, r, te, tr, infos = stats.step(a)
        te, tr = np.asarray(te, dtype=bool), np.asarray(tr, dtype=bool)
        if "episode" in infos:
            rec["ep_mask"][t] = infos["_episode"]
            rec["ep_return"][t] = infos["episode"]["r"]
            rec["ep_length"][t] = infos["episode"]["l"]
        assert np.array_equal(rec["ep_mask"][t].astype(bool), te | tr)
        rec["ep_running_return"][t] = stats.episode_returns
        rec["action"][t] = a.reshape(N)
        rec["obs"][t], rec["reward"][t], rec["terminated"][t], rec["truncated"][t] = o, r, te, tr
        if "final_observation" in infos:
            for i in range(N):
                if infos["_final_observation"][i]:
                    rec["final_obs"][t, i] = infos["final_observation"][i]
                    rec["final_mask"][t, i] = 1
        assert np.array_equal(rec["final_mask"][t].astype(bool), te | tr)
        for i, e in enumerate(venv.envs):
            rec["state_post"][t, i] = get_state(e.unwrapped)
            rec["elapsed_post"][t, i] = e._elapsed_steps
    if not save:
        venv.close()
        return dict(obs0=obs0, max_episode_steps=np.int32(limit), **rec)
    np.savez_compressed(os.path.join(HERE, f"{name}_p2_{tag}.npz"), obs0=obs0, max_episode_steps=np.int32(limit),
                        **rec)
    print(f"{name:24s} P2[{tag}]: T={T} limit={limit} term={int(rec['terminated'].sum())} "
          f"trunc={int(rec['truncated'].sum())}")
    venv.close()


if __name__ == "__main__":
    for name, (gid, S, O, nd, short) in ENVS.items():
        make_p1(name)
        make_p2(name, "default", T=450)
        make_p2(name, "short", T=200, max_episode_steps=short)
