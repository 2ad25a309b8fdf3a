"""HipEnv — the single-environment surface of the engine: gym.Env's step() / reset() contract (gym/core.py:75-184) for
`gym.make("hip/<id>")` without `num_envs`, so that the reference's README loop (README.md:29-41) runs unchanged:

    env = gym.make("hip/CartPole-v1")
    observation, info = env.reset(seed=42)
    for _ in range(1000):
        observation, reward, terminated, truncated, info = env.step(policy(observation))
        if terminated or truncated:
            observation, info = env.reset()
    env.close()

Nobody needs a GPU for ONE environment — this exists so that code written against a single env (evaluation loops, env checkers, the
determinism tests of tests/envs/test_envs.py:63-115) meets the same dynamics, TimeLimit and error behaviour as the vector envs: it is a
one-env vector engine WITHOUT autoreset (MXV_FLAG_NO_AUTORESET: a finished env stays finished until reset(), as gym.Env prescribes)
with the batch axis squeezed away.  What gym.make wraps around the reference's envs lives in the engine already:

  TimeLimit       (gym/wrappers/time_limit.py:39-68)     truncated = elapsed >= max_episode_steps, the counter restarts at reset()
  OrderEnforcing  (gym/wrappers/order_enforcing.py:33-37) step() before reset() raises ResetNeeded
  `assert self.action_space.contains(action)`            AssertionError with the reference's message (cartpole.py:131-132)
  CartPole's steps_beyond_terminated (cartpole.py:169-184): the step the pole falls pays 1.0, later steps without reset() 0.0

Classic-control ids only (the toy_text engines always autoreset: use their vector envs).  Returns unbatched observations (float32
(O,)), a Python float reward, Python bools, and an empty info dict — the types the reference's envs hand back.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import error
from .vector_env import HipVectorEnv


class Env:
    """Attribute surface of gym.Env (gym/core.py:75-226) that the adapter fills; `gym_amd.interop.as_reference_env` makes instances
    real gym.Env objects when the reference is importable."""

    metadata = {"render_modes": []}
    render_mode = None
    reward_range = (-float("inf"), float("inf"))
    spec = None

    @property
    def unwrapped(self):
        return self

    def render(self):
        return None

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()
        return False


class _OneOfTheEngine(Env):
    """What both single-env adapters share: lifecycle, gym.Env's np_random, the spec that shows the engine's TimeLimit."""

    closed = False

    def close(self):
        if not self.closed:
            self._vec.close()
            self.closed = True

    def _assert_open(self):
        if self.closed:
            raise error.ClosedEnvironmentError(f"Trying to operate on `{type(self).__name__}`, after a call to `close()`.")

    def _seed_np_random(self, seed):
        if seed is not None:      # gym.Env.reset (core.py:155-157): `self._np_random, seed = seeding.np_random(seed)`
            from .spaces import np_random

            self.__dict__["_np_random"], _ = np_random(seed)

    # gym.Env's generator (core.py:185-199): seeded by reset(seed=...), created on first use otherwise — the object callers and the
    # reference's env checker expect an env to own (gym/utils/env_checker.py:83-123).  The engine's DYNAMICS do not draw from it: reset states
    # come from the Philox4x32-10 streams keyed by the same seed, so reset(seed=s) is reproducible either way.
    @property
    def np_random(self):
        if self.__dict__.get("_np_random") is None:
            from .spaces import np_random

            self.__dict__["_np_random"], _ = np_random()
        return self.__dict__["_np_random"]

    @np_random.setter
    def np_random(self, value):
        self.__dict__["_np_random"] = value

    # `spec`: gym.make("hip/<id>") assigns the registry's EnvSpec after construction (gym/envs/registration.py:657).  The hip/ ids are
    # registered WITHOUT max_episode_steps — the limit lives in the engine, and a registered one would make gym.make put its own TimeLimit
    # wrapper on top — so the spec the env shows is a copy that carries the engine's limit: code that reads env.spec.max_episode_steps
    # (the reference's own tests do, tests/wrappers/test_record_episode_statistics.py:20) finds what the episode is actually cut at.
    @property
    def spec(self):
        return self.__dict__.get("_spec")

    @spec.setter
    def spec(self, s):
        if s is not None and hasattr(s, "max_episode_steps") and s.max_episode_steps is None and "_vec" in self.__dict__:
            limit = int(self._engine_limit() or 0)
            if limit > 0:
                import copy

                s = copy.copy(s)
                s.max_episode_steps = limit
        self.__dict__["_spec"] = s

    def __repr__(self):
        return f"<{type(self).__name__}<{self.spec.id}>>"


class HipEnv(_OneOfTheEngine):
    def __init__(self, id: str, *, device: int = 0, max_episode_steps: Optional[int] = None, render_mode=None, **kwargs):
        if render_mode is not None:
            raise NotImplementedError("the device engine has no renderer (render_mode must be None)")
        self._vec = HipVectorEnv(id, 1, device=device, max_episode_steps=max_episode_steps, autoreset=False, copy=True, **kwargs)
        self.spec = self._vec.spec
        self.observation_space = self._vec.single_observation_space
        self.action_space = self._vec.single_action_space
        self._discrete = self._vec._discrete
        self.closed = False

    # -- gym.Env --------------------------------------------------------------------------------------------------------------------
    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        """gym/core.py:129-165: reset(seed=...) reseeds the env's generator (here: the engine's Philox streams, key = seed), reset()
        continues it; `options` as the classic-control envs read them (classic_control/utils.py:17-46: {"low", "high"} / Pendulum's
        {"x_init", "y_init"})."""
        self._assert_open()
        self._seed_np_random(seed)
        obs, _ = self._vec.reset(seed=seed, options=options)
        return np.array(obs[0], dtype=np.float32), {}

    def step(self, action):
        self._assert_open()
        if not self.action_space.contains(action):       # `assert self.action_space.contains(action), err_msg` (cartpole.py:131-132)
            if not (self._discrete is False and np.shape(action) == (1,)):    # Box envs index action[0] / clip it: any (1,) array-like is accepted
                raise AssertionError(f"{action!r} ({type(action)}) invalid")
        a = np.asarray(action)
        batch = a.reshape(1).astype(np.int64) if self._discrete else a.reshape(1, 1).astype(np.float32)
        obs, rew, term, trunc, _ = self._vec.step(batch)
        return np.array(obs[0], dtype=np.float32), float(rew[0]), bool(term[0]), bool(trunc[0]), {}

    def _engine_limit(self):
        return self._vec.get_attr("_max_episode_steps")[0]

    # the attributes of the reference's env objects (env.unwrapped.gravity, .state, ...)
    def __getattr__(self, name):
        if name == "_np_random":      # (not seeded yet: None, like gym.Env's class attribute)
            return None
        if name.startswith("_"):
            raise AttributeError(name)
        vec = self.__dict__.get("_vec")
        if vec is None:
            raise AttributeError(name)
        if name == "state":
            return np.array(vec.handle.get_state()[0][:, 0])
        if name in ("_elapsed_steps", "elapsed_steps"):
            return int(vec.handle.get_state()[1][0])
        try:
            return vec.get_attr(name)[0]
        except (AttributeError, NotImplementedError):
            raise AttributeError(f"{type(self).__name__} has no attribute {name!r}") from None

    def __getstate__(self):
        return {"_vec": self._vec, "closed": self.closed}

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.spec = self._vec.spec
        self.observation_space = self._vec.single_observation_space
        self.action_space = self._vec.single_action_space
        self._discrete = self._vec._discrete


class HipToyTextEnv(_OneOfTheEngine):
    """The single-env surface over ONE env of a toy_text engine (FrozenLake / Taxi / CliffWalking tables, Blackjack): Python-int (tuple)
    observations, float reward, Python bools, the env's own info dict ({"prob": p}, Taxi's "action_mask").  The toy_text kernels always
    autoreset (there is no MXV_FLAG_NO_AUTORESET for them), so the adapter undoes it at the surface: the step that ends an episode hands
    back the TERMINAL observation and info (the engine's final_observation / final_info) and keeps the observation of the episode the
    engine has already begun; the reset() that gym.Env prescribes next returns that one — drawn from the env's own stream, exactly what a
    reset would have drawn — and reset(seed=...) reseeds and resets for real.  (Stepping on WITHOUT reset() continues in the new episode
    instead of the reference's frozen terminal state: undefined behaviour there, flagged by its PassiveEnvChecker.)"""

    def __init__(self, id: str, *, render_mode=None, **kwargs):
        if render_mode is not None:
            raise NotImplementedError("the device engine has no renderer (render_mode must be None)")
        from .vector_env import make

        self._vec = make(id, 1, **kwargs)
        self._blackjack = id == "Blackjack-v1"
        self.spec = self._vec.spec
        self.observation_space = self._vec.single_observation_space
        self.action_space = self._vec.single_action_space
        self._begun = None          # (observation, info) of the episode the engine began by itself

    def _engine_limit(self):
        return getattr(self._vec, "_limit", None)

    def _obs(self, obs):
        if self._blackjack:
            return (int(obs[0][0]), int(obs[1][0]), bool(obs[2][0]))       # blackjack.py:_get_obs
        return int(obs[0])

    @staticmethod
    def _info(infos):
        out = {}
        for k, v in infos.items():
            if not k.startswith("_") and k not in ("final_observation", "final_info"):
                x = v[0]
                out[k] = x.item() if isinstance(x, np.generic) else x
        return out

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        self._assert_open()
        self._seed_np_random(seed)
        if seed is None and self._begun is not None:
            begun, self._begun = self._begun, None
            return begun
        self._begun = None
        obs, infos = self._vec.reset(seed=seed)
        return self._obs(obs), self._info(infos)

    def step(self, action):
        self._assert_open()
        obs, rew, term, trunc, infos = self._vec.step(np.asarray([action]))
        self._begun = None
        if term[0] or trunc[0]:
            self._begun = (self._obs(obs), self._info(infos))
            fin = infos["final_observation"][0]
            last = fin if isinstance(fin, tuple) else int(fin)
            return last, float(rew[0]), bool(term[0]), bool(trunc[0]), dict(infos["final_info"][0] or {})
        return self._obs(obs), float(rew[0]), bool(term[0]), bool(trunc[0]), self._info(infos)

    def __getattr__(self, name):
        if name == "_np_random":
            return None
        if name.startswith("_"):
            raise AttributeError(name)
        vec = self.__dict__.get("_vec")
        if vec is None:
            raise AttributeError(name)
        try:
            return vec.get_attr(name)[0]
        except (AttributeError, NotImplementedError, TypeError):
            raise AttributeError(f"{type(self).__name__} has no attribute {name!r}") from None

    def __getstate__(self):
        return {"_vec": self._vec, "closed": self.closed, "_blackjack": self._blackjack, "_begun": self._begun}

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.spec = self._vec.spec
        self.observation_space = self._vec.single_observation_space
        self.action_space = self._vec.single_action_space
