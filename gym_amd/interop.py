"""Interop with the reference's own classes ("a thin Python adapter subclassing gym.vector.VectorEnv,
imports the reference lazily").

The engine's host side is gym-free: `gym_amd.vector_env.HipVectorEnv` derives from a private mirror of
`gym.vector.VectorEnv`, returns private `Box` / `Discrete` / `MultiDiscrete` objects and raises private exception types, so
the package imports and runs without `gym` installed.  Code written against the reference, however, checks identities:

    assert isinstance(env, VectorEnv)                    gym/vector/vector_env.py:289  (VectorEnvWrapper.__init__)
    isinstance(self.action_space, Box)                   gym/wrappers/clip_action.py:28
    except gym.error.ResetNeeded: ...                    user code

When `gym` is importable, `as_reference_env(env)` turns an engine object into one that passes all three — without copying
anything out of the reference and without the core importing it:

  * class: a dynamic subclass  type(cls.__name__, (cls, gym.vector.VectorEnv), {})  — the engine's methods come first in the
    MRO, `isinstance(env, gym.vector.VectorEnv)` (and `gym.Env`) hold;
  * spaces: `observation_space`, `action_space`, `single_*` are rebuilt as `gym.spaces.Box / Discrete / MultiDiscrete / Tuple`
    with the same bounds, dtypes and generator state (so seeded `sample()` streams are unchanged);
  * errors: `gym_amd.error.<Name>` is rebound to a class deriving from BOTH the engine's exception and `gym.error.<Name>`;
    handlers written against either hierarchy catch it.

`gym_amd.plugin.make_vector` — the entry point behind `gym.make("hip/<id>")` — applies it to everything it returns.
"""
from __future__ import annotations

import copy
import sys

from . import error, spaces

_ERROR_NAMES = ("Error", "UnregisteredEnv", "ResetNeeded", "InvalidAction", "AlreadyPendingCallError", "NoAsyncCallError",
                "ClosedEnvironmentError", "CustomSpaceError")
_class_cache: dict = {}
_errors_bound_to = None


def reference(required: bool = False):
    """The reference package (`gym`) if it can be imported, else None (or ImportError when `required`)."""
    mod = sys.modules.get("gym")
    if mod is not None and hasattr(mod, "vector"):
        return mod
    try:
        import gym  # noqa: PLC0415  (lazy on purpose)
        import gym.vector  # noqa: F401

        return gym
    except Exception:
        if required:
            raise
        return None


def bind_errors(gym) -> None:
    """Rebind gym_amd.error.<Name> to classes that are ALSO gym.error.<Name> (idempotent).  Exceptions are looked up on the
    module at raise time (`error.ResetNeeded(...)`), so every raise after this call produces the dual-typed class; the engine's
    original classes stay their bases, so `except` clauses that captured them earlier still match."""
    global _errors_bound_to
    if _errors_bound_to is gym:
        return
    for name in _ERROR_NAMES:
        ours = getattr(error, name)
        theirs = getattr(gym.error, name, None)
        if theirs is None or issubclass(ours, theirs):
            continue
        bases = (ours, theirs)
        setattr(error, name, type(name, bases, {"__module__": error.__name__, "__doc__": ours.__doc__}))
    _errors_bound_to = gym


def to_reference_space(space, gym):
    """The same space as a gym.spaces object (bounds, dtype, shape and generator state preserved)."""
    rs = gym.spaces
    if isinstance(space, rs.Space):
        return space
    rng = copy.deepcopy(space._np_random) if getattr(space, "_np_random", None) is not None else None
    if isinstance(space, spaces.Box):
        out = rs.Box(low=space.low, high=space.high, shape=space.shape, dtype=space.dtype.type)
        # the private Box keeps bounded_below / bounded_above from the untruncated bounds: so does the reference's constructor
    elif isinstance(space, spaces.Discrete):
        out = rs.Discrete(space.n, start=space.start)
    elif isinstance(space, spaces.MultiDiscrete):
        out = rs.MultiDiscrete(space.nvec, dtype=space.dtype.type)
    elif isinstance(space, spaces.Tuple):
        out = rs.Tuple(tuple(to_reference_space(s, gym) for s in space.spaces))
    else:
        raise TypeError(f"no gym.spaces equivalent for {type(space).__name__}")
    if rng is not None:
        out._np_random = rng
    return out


def reference_class(cls, gym):
    """Dynamic subclass of `cls` and gym.vector.VectorEnv — gym.Env for the single-env adapter (gym_amd.single_env.HipEnv) — cached per class."""
    key = (cls, id(gym))
    if key not in _class_cache:
        from .single_env import Env as _SingleEnv

        base = gym.Env if issubclass(cls, _SingleEnv) else gym.vector.VectorEnv
        if issubclass(cls, base):
            _class_cache[key] = cls
        else:
            ns = {"__module__": cls.__module__, "__doc__": cls.__doc__, "_reference_base": base,
                  # gym.Env defines `unwrapped` / `np_random` / `__enter__`... behind the engine's classes in the MRO; the
                  # engine's own definitions (vector env = its own base env) keep winning because `cls` comes first.
                  "__reduce_ex__": _reduce_reference_env}
            _class_cache[key] = type(cls.__name__, (cls, base), ns)
    return _class_cache[key]


def _rebuild_reference_env(cls_module, cls_name, state):
    import importlib

    cls = getattr(importlib.import_module(cls_module), cls_name)
    env = cls.__new__(cls)
    if hasattr(env, "__setstate__"):
        env.__setstate__(state)
    else:
        env.__dict__.update(state)
    gym = reference()
    return as_reference_env(env, gym) if gym is not None else env


def _reduce_reference_env(self, protocol):
    """Pickle as the engine's own (importable) class; unpickling re-applies the interop when gym is importable there."""
    plain = type(self).__mro__[1]
    state = self.__getstate__() if hasattr(self, "__getstate__") else dict(self.__dict__)
    return _rebuild_reference_env, (plain.__module__, plain.__name__, state)


def as_reference_env(env, gym=None):
    """Make `env` (HipVectorEnv / HipTabularVectorEnv / HipBlackjackVectorEnv) an instance of gym.vector.VectorEnv — a HipEnv an
    instance of gym.Env — with gym.spaces spaces and gym.error exceptions.  Returns `env` (modified in place); a no-op without gym."""
    gym = gym or reference()
    if gym is None:
        return env
    bind_errors(gym)
    env.__class__ = reference_class(type(env), gym)
    for name in ("observation_space", "action_space", "single_observation_space", "single_action_space"):
        sp = getattr(env, name, None)
        if sp is not None:
            setattr(env, name, to_reference_space(sp, gym))
    return env


def is_reference_env(env) -> bool:
    gym = reference()
    return gym is not None and isinstance(env, (gym.vector.VectorEnv, gym.Env))
