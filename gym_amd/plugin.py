"""Registration of the engine with the reference's registry (SURVEY.md §8b extension points (ii)-(iv), §8f-3).

Three ways in, none of which touches the reference tree:

  * explicit:      `import gym_amd.plugin; gym_amd.plugin.register_envs()`  then  `gym.make("hip/CartPole-v1", num_envs=4096)`
  * import hook:   `gym.make("gym_amd.plugin:hip/CartPole-v1", num_envs=4096)` — gym imports this module first
                   (gym/envs/registration.py:537-545) and importing it registers the ids
  * entry point:   an installed distribution declares  [project.entry-points."gym.envs"]  hip = "gym_amd.plugin:register_envs"
                   (pyproject.toml) and gym loads it at import (registration.py:266-309, gym/envs/__init__.py:5)

`gym.make("hip/<id>")` WITHOUT num_envs returns one environment with gym.Env's own contract (gym_amd.single_env.HipEnv / HipToyTextEnv:
unbatched observations, float reward, bool flags, no autoreset at the surface — the reference's README loop runs unchanged); with num_envs,
a vector env.

The registered ids are `hip/<reference id>` for every id of gym_amd.registration.registry (classic control) and of
gym_amd.toy_text.TOY_TEXT_REGISTRY (FrozenLake / Taxi / CliffWalking) plus Blackjack-v1.  Their entry point returns a HipVectorEnv /
HipTabularVectorEnv / HipBlackjackVectorEnv — a *vector* env — so the specs switch off everything gym.make would wrap around a single env:
`order_enforce=False`, `disable_env_checker=True`, `max_episode_steps=None` (TimeLimit lives inside the kernels; pass
`time_limit=` to change it, not gym.make's own `max_episode_steps=` which would add the single-env wrapper).
"""
from __future__ import annotations

from .registration import registry

NAMESPACE = "hip"


def make_vector(id: str, num_envs=None, time_limit=None, **kwargs):
    """Entry point of the registered specs.  With `num_envs`: a vector env, an instance of the reference's gym.vector.VectorEnv; without
    it (`gym.make("hip/CartPole-v1")`, as the reference's README writes it): ONE environment with gym.Env's unbatched step() / reset()
    contract (gym_amd.single_env.HipEnv for the classic-control ids, HipToyTextEnv for the toy_text ones).  Either way with gym.spaces spaces and gym.error exceptions
    (gym_amd.interop) — gym.make is calling, so gym is importable."""
    from .interop import as_reference_env

    if time_limit is not None:
        kwargs["max_episode_steps"] = time_limit
    if num_envs is None:
        from .registration import registry as classic
        from .single_env import HipEnv, HipToyTextEnv

        return as_reference_env(HipEnv(id, **kwargs) if id in classic else HipToyTextEnv(id, **kwargs))
    from .vector_env import make

    return as_reference_env(make(id, num_envs, **kwargs))


def register_envs(gym_module=None) -> list:
    """Register `hip/<id>` for every supported id; returns the registered ids.  Safe to call twice."""
    if gym_module is None:
        import gym as gym_module  # the reference (or its successor) must be importable for this entry point
    from .toy_text import TOY_TEXT_REGISTRY

    done = []
    thresholds = [(env_id, spec.reward_threshold) for env_id, spec in list(registry.items()) + list(TOY_TEXT_REGISTRY.items())]
    thresholds.append(("Blackjack-v1", None))  # gym/envs/__init__.py:95-99
    for env_id, reward_threshold in thresholds:
        full = f"{NAMESPACE}/{env_id}"
        if full not in gym_module.envs.registry:
            gym_module.register(id=full, entry_point="gym_amd.plugin:make_vector", reward_threshold=reward_threshold,
                                max_episode_steps=None, order_enforce=False, disable_env_checker=True,
                                kwargs={"id": env_id})
        done.append(full)
    return done


try:  # the import hook path: importing this module is enough
    import gym as _gym

    register_envs(_gym)
except Exception:  # gym not importable (the engine works without it)
    pass
