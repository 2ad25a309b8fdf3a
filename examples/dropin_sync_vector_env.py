#!/usr/bin/env python3
"""The drop-in, side by side: BASELINE.json configs[0] — `gym.vector.make("CartPole-v1", num_envs=8, asynchronous=False)`, seeded reset,
1000 random steps — written once against the `gym.vector` interface and run on whatever `vector_make` is handed in: the reference's
`gym.vector.make` (when `gym` is importable, e.g. PYTHONPATH=/root/reference) and this engine's `gym_amd.vector.make`.

    python examples/dropin_sync_vector_env.py [--envs 8] [--steps 1000] [--id CartPole-v1]

The loop body does not know which one it is driving: observations float32 (N, O), rewards float64 (N,), terminated / truncated bool
(N,), `infos["final_observation"]` / `infos["_final_observation"]` on the steps that end an episode (the returned observation is then the
first one of the next episode: gym/vector/sync_vector_env.py:152-156).  The two runs draw different random numbers (PCG64 per sub-env
there, Philox streams here), so trajectories differ; contracts, dtypes and statistics (mean episode length ~22 for a random CartPole
policy) do not.  What changes is where the time goes: 8 Python envs stepped in a loop vs one kernel launch — and that the second one can
be asked for 2^20 envs."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ppo_recipe(wrappers):
    """The wrapper list of the continuous-control PPO scripts, from whichever module provides the classes (gym.wrappers for the reference,
    gym_amd.wrappers for the engine — gym_amd.make also accepts gym.wrappers' own): every sub-env clips its actions, normalises and clips
    its observations with ITS OWN running statistics, normalises and clips its rewards."""
    import functools

    return [wrappers.ClipAction, wrappers.NormalizeObservation, functools.partial(wrappers.TransformObservation, f=lambda obs: np.clip(obs, -10, 10)),
            functools.partial(wrappers.NormalizeReward, gamma=0.99), functools.partial(wrappers.TransformReward, f=lambda reward: np.clip(reward, -10, 10))]


def run(vector_make, env_id, num_envs, steps, wrappers=None):
    env = vector_make(env_id, num_envs=num_envs, asynchronous=False, **({} if wrappers is None else {"wrappers": wrappers}))
    obs, infos = env.reset(seed=0)
    env.action_space.seed(0)
    assert obs.shape == (num_envs,) + env.single_observation_space.shape and obs.dtype == env.single_observation_space.dtype
    episodes, lengths, current = 0, 0, np.zeros(num_envs, dtype=np.int64)
    t0 = time.perf_counter()
    for _ in range(steps):
        obs, rewards, terminated, truncated, infos = env.step(env.action_space.sample())
        current += 1
        done = terminated | truncated
        if done.any():
            assert np.array_equal(infos["_final_observation"], done)          # the terminal observation travels in the infos
            episodes += int(done.sum())
            lengths += int(current[done].sum())
            current[done] = 0
    dt = time.perf_counter() - t0
    assert rewards.dtype == np.float64 and terminated.dtype == np.bool_ and truncated.dtype == np.bool_
    env.close()
    return {"env_steps_per_s": num_envs * steps / dt, "us_per_vector_step": dt / steps * 1e6, "episodes": episodes,
            "mean_episode_length": lengths / max(episodes, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--id", default="CartPole-v1")
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--recipe", action="store_true", help="wrappers=[ClipAction, NormalizeObservation, clip, NormalizeReward, clip] around every "
                                                           "sub-env (Box-action ids: --id Pendulum-v1)")
    args = ap.parse_args()

    try:
        for name, val in (("bool8", np.bool_), ("float_", np.float64)):      # NumPy-2 aliases the reference still uses
            if not hasattr(np, name):
                setattr(np, name, val)
        import gym

        gym.logger.set_level(gym.logger.ERROR)
        r = run(lambda id, **kw: gym.vector.make(id, disable_env_checker=True, **kw), args.id, args.envs, args.steps,
                ppo_recipe(gym.wrappers) if args.recipe else None)
        print(f"gym.vector.make      {args.id} x{args.envs}: {r['env_steps_per_s']:12.0f} env-steps/s  ({r['us_per_vector_step']:8.1f} us per vector step, "
              f"{r['episodes']} episodes, mean length {r['mean_episode_length']:.1f})")
    except ImportError:
        print("gym is not importable here (PYTHONPATH=/root/reference): skipping the reference run")

    import gym_amd

    recipe = ppo_recipe(gym_amd.wrappers) if args.recipe else None
    r = run(gym_amd.vector.make, args.id, args.envs, args.steps, recipe)
    print(f"gym_amd.vector.make  {args.id} x{args.envs}: {r['env_steps_per_s']:12.0f} env-steps/s  ({r['us_per_vector_step']:8.1f} us per vector step, "
          f"{r['episodes']} episodes, mean length {r['mean_episode_length']:.1f})")
    big = 1 << 20
    r = run(gym_amd.vector.make, args.id, big, 50, recipe)
    print(f"gym_amd.vector.make  {args.id} x{big}: {r['env_steps_per_s']:12.0f} env-steps/s  ({r['us_per_vector_step']:8.1f} us per vector step: NumPy "
          "arrays over PCIe both ways; DeviceRollout keeps them on the device)")


if __name__ == "__main__":
    main()
