"""HipVectorEnv — drop-in for gym.vector.SyncVectorEnv on the classic-control env ids.

Host-side mirror of the reference's vector API: gym.vector.VectorEnv
(gym/vector/vector_env.py:12-274) for attributes/method names, gym.vector.SyncVectorEnv
(gym/vector/sync_vector_env.py:15-236) for behaviour (dtypes, autoreset, infos, errors).
All arithmetic happens in the HIP engine behind include/mxv.h; this file only converts
between the NumPy contract of the reference and the engine's buffers.
"""
from __future__ import annotations

import os
from typing import List, Optional, Union

import numpy as np

from . import _native, error
from .registration import (ACROBOT as ACROBOT_KIND, CARTPOLE as CARTPOLE_KIND, CTOR_KWARGS, ENUM_PARAMS, MOUNTAINCAR as MOUNTAINCAR_KIND,
                           MOUNTAINCAR_CONT as MOUNTAINCAR_CONT_KIND, PARAM_NAMES, PENDULUM, single_spaces, spec as _spec)

# names VectorEnv.call() answers besides the physics attributes (read-only; no arguments)
READ_ONLY_CALLS = ("state", "_elapsed_steps", "_max_episode_steps", "spec", "render_mode", "action_space", "observation_space")
from .spaces import Discrete, batch_space

__all__ = ["VectorEnv", "HipVectorEnv", "make"]


def _verify_number_and_cast(x) -> float:
    """gym/envs/classic_control/utils.py:8-14."""
    try:
        return float(x)
    except (ValueError, TypeError):
        raise ValueError(f"An option ({x}) could not be converted to a float.")


class _Pending:
    __slots__ = ("build",)

    def __init__(self, build):
        self.build = build


class LazyInfos(dict):
    """infos dict whose `final_observation` / `final_info` object arrays (vector_env.py:208-258:
    length-N object ndarrays with None holes, plus `_key` boolean masks) are materialised on first
    access — building them eagerly is a Python loop over every finished env of every step.

    Every way of getting a value out resolves the placeholder first, so callers only ever see what the reference's plain
    dict holds: item access, get / pop / popitem / setdefault, items / values, iteration-based copies (`dict(infos)`,
    `{**infos}`, `infos | other` — defining `__iter__` takes CPython's dict_merge off its raw-storage fast path, it then goes
    through keys() + __getitem__), copy() (returns a plain dict), repr, ==, pickle and copy.copy / copy.deepcopy (reduce to
    a plain dict)."""

    def _resolve(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(v, _Pending):
            v = v.build()
            dict.__setitem__(self, key, v)
        return v

    def _resolve_all(self):
        for k in list(dict.keys(self)):
            self._resolve(k)
        return self

    def __getitem__(self, key):
        return self._resolve(key)

    def __iter__(self):
        return iter(list(dict.keys(self)))

    def get(self, key, default=None):
        return self._resolve(key) if key in self else default

    def items(self):
        return [(k, self._resolve(k)) for k in dict.keys(self)]

    def values(self):
        return [self._resolve(k) for k in dict.keys(self)]

    def pop(self, key, *default):
        if key in self:
            self._resolve(key)
        return dict.pop(self, key, *default)

    def popitem(self):
        self._resolve_all()
        return dict.popitem(self)

    def setdefault(self, key, default=None):
        if key in self:
            return self._resolve(key)
        return dict.setdefault(self, key, default)

    def copy(self):
        return dict(self.items())

    __copy__ = copy

    def __deepcopy__(self, memo):
        import copy as _copy

        return _copy.deepcopy(self.copy(), memo)

    def __reduce_ex__(self, protocol):
        return (dict, (self.copy(),))

    def __reduce__(self):
        return (dict, (self.copy(),))

    def __or__(self, other):
        return self.copy() | (other.copy() if isinstance(other, LazyInfos) else other)

    def __ror__(self, other):
        return (other.copy() if isinstance(other, LazyInfos) else other) | self.copy()

    def __ior__(self, other):
        self.update(other)
        return self

    def __repr__(self):
        return repr(self.copy())

    def __eq__(self, other):
        return self.copy() == (dict(other.items()) if isinstance(other, dict) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class VectorEnv:
    """Attribute/method surface of gym.vector.VectorEnv (vector_env.py:25-206)."""

    def __init__(self, num_envs: int, observation_space, action_space):
        self.num_envs = num_envs
        self.is_vector_env = True
        self.observation_space = batch_space(observation_space, n=num_envs)
        self.action_space = batch_space(action_space, n=num_envs)
        self.closed = False
        self.viewer = None
        self.single_observation_space = observation_space
        self.single_action_space = action_space

    def reset_async(self, seed=None, options=None):
        pass

    def reset_wait(self, seed=None, options=None):
        raise NotImplementedError("VectorEnv does not implement function")

    def reset(self, *, seed: Optional[Union[int, List[int]]] = None, options: Optional[dict] = None):
        self.reset_async(seed=seed, options=options)
        return self.reset_wait(seed=seed, options=options)

    def step_async(self, actions):
        pass

    def step_wait(self, **kwargs):
        raise NotImplementedError()

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def call_async(self, name, *args, **kwargs):
        pass

    def call_wait(self, **kwargs):
        pass

    def call(self, name: str, *args, **kwargs):
        self.call_async(name, *args, **kwargs)
        return self.call_wait()

    def get_attr(self, name: str):
        return self.call(name)

    def set_attr(self, name: str, values):
        pass

    def close_extras(self, **kwargs):
        pass

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        self.closed = True

    def __del__(self):
        if not getattr(self, "closed", True):
            self.close()

    @property
    def unwrapped(self):
        """gym.Env.unwrapped (gym/core.py:186-193): a vector env is its own base env."""
        return self

    def __repr__(self) -> str:
        spec_ = getattr(self, "spec", None)
        if spec_ is None:
            return f"{self.__class__.__name__}({self.num_envs})"
        return f"{self.__class__.__name__}({spec_.id}, {self.num_envs})"


class VectorEnvWrapper(VectorEnv):
    """gym.vector.VectorEnvWrapper (gym/vector/vector_env.py:277-332): base class of user wrappers around a vector env.
    The VectorEnv methods are forwarded explicitly, every other public attribute implicitly; a subclass overrides
    `step_wait` / `reset_wait` (as the reference's own test wrappers do) and `step()` / `reset()` pick that up."""

    def __init__(self, env: VectorEnv):
        assert isinstance(env, VectorEnv)
        self.env = env

    def reset_async(self, **kwargs):
        return self.env.reset_async(**kwargs)

    def reset_wait(self, **kwargs):
        return self.env.reset_wait(**kwargs)

    def step_async(self, actions):
        return self.env.step_async(actions)

    def step_wait(self):
        return self.env.step_wait()

    def close(self, **kwargs):
        return self.env.close(**kwargs)

    def close_extras(self, **kwargs):
        return self.env.close_extras(**kwargs)

    def call(self, name, *args, **kwargs):
        return self.env.call(name, *args, **kwargs)

    def set_attr(self, name, values):
        return self.env.set_attr(name, values)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(f"attempted to get missing private attribute '{name}'")
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __repr__(self):
        return f"<{self.__class__.__name__}, {self.env}>"

    def __del__(self):
        env = self.__dict__.get("env")
        if env is not None:
            env.__del__()


class HipVectorEnv(VectorEnv):
    """`num_envs` copies of one classic-control env, resident on one MI355X.

    Same call surface and return contract as SyncVectorEnv: observations float32 (N, O) (a fresh array
    per call), rewards float64 (N,), terminated/truncated np.bool_ (N,), infos with `final_observation`
    / `final_info` and their `_` masks when an env finished (sync_vector_env.py:152-169).

    Difference that cannot be avoided: resets draw from the engine's Philox4x32-10 streams, not from
    PCG64, so seeded initial states differ from the reference's (RNG).
    """

    metadata = {"render_modes": []}
    render_mode = None

    def __init__(self, id: str, num_envs: int = 1, *, device: int = 0, max_episode_steps: Optional[int] = None,
                 env_offset: int = 0, copy: bool = True, zero_copy: bool = False, autoreset: bool = True, **kwargs):
        self.spec = _spec(id)
        self.kind = self.spec.kind
        observation_space, action_space = single_spaces(self.kind)
        super().__init__(num_envs=num_envs, observation_space=observation_space, action_space=action_space)
        # copy=False is SyncVectorEnv's (sync_vector_env.py:61-63,163): reset/step return the internal observation buffer —
        # here a view of the engine's pinned, device-mapped I/O block that the kernel writes over PCIe (one launch and one
        # synchronisation per step, no staging copies); rewards / flags are still fresh copies, as in the reference.
        # zero_copy=True additionally returns rewards / terminated / truncated as views of that block.
        # Both are what small vector envs gain from (20 vs 28 us per step at 8 envs).  LARGE ones (step I/O above 2 MiB: the
        # kernel no longer writes host memory itself) take the copy=True path whatever was asked: it is one DMA into a pooled
        # pinned block whose views the caller owns, which is both faster than DMA-ing into one shared block and copying out of
        # it (0.90 vs 1.12-1.48 ms per step at 2^20 CartPole envs) and a stronger guarantee (nothing is overwritten).
        self.copy = copy and not zero_copy
        self.zero_copy = zero_copy
        self._views = None
        limit = self.spec.max_episode_steps if max_episode_steps is None else max_episode_steps
        self._max_episode_steps = -1 if limit is None else int(limit)
        self._discrete = isinstance(action_space, Discrete)
        entropy = int.from_bytes(os.urandom(8), "little")  # Env.reset(seed=None) = fresh OS entropy (seeding.py:24)
        # autoreset=False (MXV_FLAG_NO_AUTORESET): dynamics + TimeLimit only, a finished env stays finished until reset() — the single-env
        # contract of gym.Env (gym/core.py:75-184), which gym_amd.single_env.HipEnv builds on
        self.autoreset = bool(autoreset)
        self._handle = _native.Handle(self.kind, num_envs, self._max_episode_steps, device=device,
                                      env_offset=env_offset, seed=entropy, action_seed=entropy ^ 0x9E3779B97F4A7C15,
                                      **({} if autoreset else {"flags": _native.FLAG_NO_AUTORESET}))
        self._actions = None
        self._was_reset = False
        self._per_env = False  # True while sub-envs hold differing physics attributes (set_attr with a list)
        # large envs: info["final_observation"] rows arrive packed (indices + rows of the finished envs), never as a dense array
        self._packed = bool(getattr(self._handle, "final_packed", lambda on: False)(True))
        self._own_arrays = self.copy or self._packed     # step/reset return arrays nobody else writes to
        allowed = CTOR_KWARGS.get(self.kind, {})
        for k, v in kwargs.items():
            if k == "render_mode" and v is None:
                continue
            if k not in allowed:
                raise TypeError(f"{id} got an unexpected keyword argument {k!r}")
            self._set_param(allowed[k], v)

    # -- parameters (get_attr / set_attr / call) ---------------------------------------------------
    def _param_index(self, name: str) -> int:
        try:
            return PARAM_NAMES[self.kind].index(name)
        except ValueError:
            raise AttributeError(f"{self.spec.id} sub-environments have no attribute {name!r}") from None

    def _decode(self, name, value):
        enum = ENUM_PARAMS.get((self.kind, name))
        return enum[int(value)] if enum else value

    def _encode(self, name, value):
        enum = ENUM_PARAMS.get((self.kind, name))
        if enum:
            if value not in enum:
                raise ValueError(f"{name} must be one of {enum}, got {value!r}")
            return float(enum.index(value))
        return float(value)

    def _set_param(self, name, value):
        p = self._handle.get_params()
        p[self._param_index(name)] = self._encode(name, value)
        try:
            self._handle.set_params(p)
        except _native.MxvError as e:
            if e.code == _native.ERR_UNSUPPORTED:
                raise NotImplementedError(e.message) from None
            raise

    def call(self, name: str, *args, **kwargs) -> tuple:
        """sync_vector_env.py:171-190: attribute value (or method result) of every sub-env, as a tuple."""
        self._assert_is_running()
        if name in READ_ONLY_CALLS and not args and not kwargs:
            return self._read_only_call(name)
        if args or kwargs or name in ("step", "reset", "render", "close", "seed"):
            raise NotImplementedError(
                f"call({name!r}, ...): sub-environment METHODS are not callable on the device engine (there are no Python sub-envs); "
                f"attributes are: {sorted(PARAM_NAMES[self.kind])} (get / set) and {sorted(READ_ONLY_CALLS)} (read-only) — INTEGRATION.md")
        idx = self._param_index(name)
        if self._per_env:
            return tuple(self._decode(name, v) for v in self._handle.get_params_per_env()[idx].tolist())
        value = self._decode(name, self._handle.get_params()[idx])
        return (value,) * self.num_envs

    def _read_only_call(self, name: str) -> tuple:
        """What the reference's wrapped sub-envs answer for these names (TimeLimit / OrderEnforcing forward unknown attributes to the raw
        env): `state` — the env's fp64 state (cartpole.py:160 a tuple, the others arrays; MountainCarContinuous float32, :171);
        `_elapsed_steps`, `_max_episode_steps` (time_limit.py:43-44); `spec`, `render_mode`, `action_space`, `observation_space`."""
        n = self.num_envs
        if name == "state":
            # the container is the reference's own, which depends on whether the sub-env was just reset (reset() stores what
            # np_random.uniform returned) or stepped (step() stores what it computed): cartpole.py:202 ndarray / :160 tuple of floats;
            # mountain_car.py:160 ndarray / :147 tuple; acrobot.py:188-190 float32 ndarray / :209-218 float64 ndarray;
            # continuous_mountain_car.py:182 float64 ndarray / :171 float32 ndarray; pendulum.py:153,135 float64 ndarray both times
            st, el = self._handle.get_state()
            out = []
            for i in range(n):
                fresh, v = el[i] == 0, st[:, i]
                if self.kind in (CARTPOLE_KIND, MOUNTAINCAR_KIND) and not fresh:
                    out.append(tuple(float(x) for x in v))
                elif (self.kind == ACROBOT_KIND and fresh) or (self.kind == MOUNTAINCAR_CONT_KIND and not fresh):
                    out.append(v.astype(np.float32))
                else:
                    out.append(v.astype(np.float64))
            return tuple(out)
        if name == "_elapsed_steps":
            return tuple(int(v) for v in self._handle.get_state()[1])
        if name == "_max_episode_steps":
            return (self._handle.max_episode_steps if self._handle.max_episode_steps > 0 else None,) * n
        if name == "spec":
            return (self.spec,) * n
        if name == "render_mode":
            return (None,) * n
        if name == "action_space":
            return (self.single_action_space,) * n
        return (self.single_observation_space,) * n

    def set_attr(self, name: str, values):
        """sync_vector_env.py:192-214: list/tuple of per-env values or one broadcast value."""
        self._assert_is_running()
        if not isinstance(values, (list, tuple)):
            values = [values for _ in range(self.num_envs)]
        if len(values) != self.num_envs:
            raise ValueError("Values must be a list or tuple with length equal to the number of environments. "
                             f"Got `{len(values)}` values for {self.num_envs} environments.")
        first = values[0]
        if not self._per_env and all(v == first for v in values):
            self._set_param(name, first)   # one value for all: constants stay kernel arguments
            return
        # differing values (tests/vector/test_sync_vector_env.py:101-110): per-env parameter table on the device
        idx = self._param_index(name)
        table = self._handle.get_params_per_env()
        table[idx] = [self._encode(name, v) for v in values]
        if all(np.all(row == row[0]) for row in table):  # back to common values: leave per-env mode
            self._handle.set_params(table[:, 0].copy())
            self._per_env = False
            return
        try:
            self._handle.set_params_per_env(table)
        except _native.MxvError as e:
            if e.code == _native.ERR_UNSUPPORTED:
                raise NotImplementedError(e.message) from None
            raise
        self._per_env = True

    # -- reset ---------------------------------------------------------------------------------------
    def _reset_bounds(self, options: Optional[dict]):
        if options is None:
            return None
        if self.kind == PENDULUM:  # pendulum.py:141-152
            d = _native.default_reset_bounds(self.kind)
            x = _verify_number_and_cast(options.get("x_init")) if "x_init" in options else d[0]
            y = _verify_number_and_cast(options.get("y_init")) if "y_init" in options else d[1]
            return np.array([x, y], dtype=np.float64)
        d = _native.default_reset_bounds(self.kind)  # classic_control/utils.py:17-46
        low = _verify_number_and_cast(options.get("low")) if "low" in options else d[0]
        high = _verify_number_and_cast(options.get("high")) if "high" in options else d[1]
        if low > high:
            raise ValueError(f"Lower bound ({low}) must be lower than higher bound ({high}).")
        return np.array([low, high], dtype=np.float64)

    def reset_wait(self, seed: Optional[Union[int, List[int]]] = None, options: Optional[dict] = None):
        """sync_vector_env.py:90-129."""
        self._assert_is_running()
        bounds = self._reset_bounds(options)
        if seed is not None:
            if isinstance(seed, (int, np.integer)):
                if seed < 0:
                    raise error.Error(f"Seed must be a non-negative integer or omitted, not {seed}")
                self._handle.seed(int(seed) + 0, None)  # env i gets seed + i (sync_vector_env.py:106-107)
            else:
                seeds = list(seed)
                assert len(seeds) == self.num_envs
                for s in seeds:
                    if not (isinstance(s, (int, np.integer)) and s >= 0):
                        raise error.Error(f"Seed must be a non-negative integer or omitted, not {s}")
                self._handle.seed(0, np.array(seeds, dtype=np.uint64))
        if self._own_arrays:
            obs = self._handle.reset_host(bounds=bounds)
        else:
            io = self._io()   # creates the mapped block on first use
            self._handle.reset_mapped(bounds=bounds)
            obs = io["obs"]
        self._was_reset = True
        self._actions = None
        return obs, {}

    def _io(self):
        if self._views is None:
            self._views = self._handle.host_io()
        return self._views

    # -- step ----------------------------------------------------------------------------------------
    def step_async(self, actions):
        """sync_vector_env.py:131-133."""
        self._assert_is_running()
        if self._actions is not None:
            raise error.AlreadyPendingCallError("Calling `step_async` while waiting for a pending call to `step` to "
                                                "complete.", "step")
        n = self.num_envs
        a = np.asarray(actions)
        if self._discrete:
            if a.shape != (n,) or not np.issubdtype(a.dtype, np.integer):
                # Discrete.contains (discrete.py:83-94) accepts integers only: the reference asserts
                raise AssertionError(f"{actions!r} ({type(actions)}) invalid")
            self._actions = np.ascontiguousarray(a, dtype=self._handle.action_dtype)
        else:
            if a.size != n:
                raise AssertionError(f"expected {n} actions of shape (1,), got array of shape {a.shape}")
            self._actions = np.ascontiguousarray(a, dtype=np.float32).reshape(n)

    def step_wait(self):
        """sync_vector_env.py:135-169."""
        self._assert_is_running()
        if self._actions is None:
            raise error.NoAsyncCallError("Calling `step_wait` without any prior call to `step_async`.", "step")
        actions, self._actions = self._actions, None
        if not self._was_reset:
            raise error.ResetNeeded("Cannot call env.step() before calling env.reset()")
        try:
            if self._own_arrays:
                obs, rew, term, trunc, fin = self._handle.step_host_block(actions, want_final=True)
            else:
                io = self._io()
                io["actions"][:] = actions
                self._handle.step_mapped()
                obs, fin = io["obs"], io["final_obs"]
                rew, term, trunc = io["reward"], io["terminated"], io["truncated"]
                if not self.zero_copy:      # fresh-to-the-caller copies out of a recycling pool (no page faults per step)
                    pool = self.__dict__.setdefault("_copy_pool", _native._ArrayPool())
                    outs = []
                    for src in (rew, term, trunc):
                        dst = pool.take(src.shape, src.dtype)
                        np.copyto(dst, src)
                        outs.append(dst)
                    rew, term, trunc = outs
        except _native.MxvError as e:
            if e.code == _native.ERR_INVALID_ACTION:
                raise AssertionError(f"{actions!r} ({type(actions)}) invalid") from None
            if e.code == _native.ERR_RESET_NEEDED:
                raise error.ResetNeeded("Cannot call env.step() before calling env.reset()") from None
            raise
        infos = LazyInfos()
        done = term | trunc
        if done.any():
            n = self.num_envs
            # Everything below is deferred to first access: the index list of the finished envs (np.flatnonzero over N flags)
            # and the object arrays cost 0.5 ms per step at 2^20 envs, and most training loops never look at them.
            state = {}

            if self._packed:            # (index, row) pairs packed by the device; the library's buffer is overwritten next step
                state["idx"], state["rows"] = self._handle.final_packed_rows()

            def finished():
                if "idx" not in state:
                    state["idx"] = np.flatnonzero(done)
                    # rows of the finished envs only, copied out of the step's buffer (a pooled or shared array)
                    state["rows"] = np.take(fin, state["idx"], axis=0)
                return state["idx"], state["rows"]

            if not self._own_arrays:
                finished()          # `fin` is a view of the shared I/O block the next step overwrites: take the rows now
                done = done.copy()

            def build_final_obs():
                idx, rows = finished()
                arr = np.full(n, None, dtype=object)
                for j, i in enumerate(idx):
                    arr[i] = rows[j]
                return arr

            def build_final_info():
                idx, _ = finished()
                arr = np.full(n, None, dtype=object)
                for i in idx:
                    arr[i] = {}
                return arr

            dict.__setitem__(infos, "final_observation", _Pending(build_final_obs))
            dict.__setitem__(infos, "_final_observation", done)
            dict.__setitem__(infos, "final_info", _Pending(build_final_info))
            dict.__setitem__(infos, "_final_info", _Pending(done.copy))   # its own array, like the reference's (made on access)
        return obs, rew, term, trunc, infos

    # -- misc ----------------------------------------------------------------------------------------
    def close_extras(self, **kwargs):
        h = getattr(self, "_handle", None)
        self._views = None   # arrays the caller still holds keep the pinned block (and with it the handle) alive
        if h is not None:
            h.close()

    def _assert_is_running(self):
        if self.closed:
            raise error.ClosedEnvironmentError(f"Trying to operate on `{type(self).__name__}`, after a call to `close()`.")

    @property
    def unwrapped(self):
        """gym.Env surface used by gym.make (`env.unwrapped.spec = ...`, gym/envs/registration.py:656)."""
        return self

    # -- pickling: how the reference's envs are checkpointed (tests/envs/test_envs.py:192-200) ----------------------
    def __getstate__(self):
        self._assert_is_running()
        d = {k: v for k, v in self.__dict__.items() if k not in ("_handle", "_views", "_copy_pool")}
        d["_snapshot"] = self._handle.snapshot()
        d["_device"] = self._handle.device
        return d

    def __setstate__(self, d):
        d = dict(d)
        snap, device = d.pop("_snapshot"), d.pop("_device")
        self.__dict__.update(d)
        self._views = None
        self._handle = _native.Handle(snap["env_id"], snap["num_envs"], snap["max_episode_steps"], device=device,
                                      env_offset=snap["env_offset"], seed=snap["base_seed"], action_seed=snap["action_seed"],
                                      flags=snap["flags"])
        self._handle.restore(snap)
        self._packed = bool(self._handle.final_packed(True))
        self._own_arrays = self.copy or self._packed

    # -- escape hatch for device-resident use ----------------------------------------------------------
    @property
    def handle(self) -> "_native.Handle":
        """The engine handle (device-pointer API: see gym_amd.rollout.DeviceRollout)."""
        return self._handle


def _sub_env_wrappers(wrappers):
    """gym.vector.make(..., wrappers=...) applies callables to every Python sub-env (gym/vector/__init__.py:56-65).  There are no Python
    sub-envs here; the wrappers that are pure functions of what the engine already owns are recognised — passed as the class or as a
    functools.partial of it, the reference's class or this package's — and mapped to the engine's equivalent with the SAME per-sub-env
    semantics:
        TimeLimit(max_episode_steps=k)        the engine's TimeLimit counter with min(k, the id's own limit)  (time_limit.py:50-53: the outer
                                              wrapper truncates at k, the registered inner one at the spec's limit)
        RecordEpisodeStatistics(deque_size=)  the fused episode accumulators, reported in infos["final_info"][i]["episode"]
        OrderEnforcing, PassiveEnvChecker     what gym.make applies anyway: accepted, nothing to do
        ClipAction                            Box-action ids only (clip_action.py:28 asserts it): np.clip(action, low, high).  Pendulum-v1: the
                                              step's first operation is the same clip and nothing else sees the action (pendulum.py:126-129):
                                              a no-op.  MountainCarContinuous-v0: the reward uses the action AS GIVEN
                                              (continuous_mountain_car.py:169), so the clip is applied for real (SubEnvClipAction)
        RescaleAction(min_action=, max_action=)
                                              Box-action ids (rescale_action.py:45): the action space becomes Box(min, max), actions are
                                              mapped affinely onto the env's own bounds and clipped (SubEnvRescaleAction)
        TransformObservation(f=), TransformReward(f=)
                                              f on every sub-env's observation / reward (transform_observation.py:34-43,
                                              transform_reward.py:36-44): on the host, over the whole batch once f(batch) has been
                                              checked against f(row) on the first batch (wrappers._RowMap), row by row otherwise
        FlattenObservation                    classic-control observations are flat Box vectors already (flatten_observation.py:33-43): a no-op
        NormalizeObservation(epsilon=), NormalizeReward(gamma=, epsilon=)
                                              per-sub-env running statistics (a batch of one per update) — a DIFFERENT normalisation from
                                              the vector-level gym_amd.NormalizeObservation / NormalizeReward (batch statistics): mapped
                                              to SubEnvNormalizeObservation / SubEnvNormalizeReward — the mxv_subnorm_* device kernels
                                              from 4096 sub-envs, the same arithmetic in NumPy over the adapter's arrays below
    in the order given (innermost first, as the reference applies them).  Anything else (lambdas that wrap the env themselves, pixel /
    frame-stack transforms, ...) cannot run inside the device engine.  Returns (max_episode_steps or None, [post-construction vector wrappers])."""
    import functools

    if wrappers is None:
        return None, []
    if callable(wrappers):
        wrappers = [wrappers]
    try:
        wrappers = list(wrappers)
    except TypeError:
        raise NotImplementedError("`wrappers` must be a callable or an iterable of callables (gym/vector/__init__.py:56-65)") from None
    limit, post = None, []
    for w in wrappers:
        fn, args, kw = (w.func, w.args, dict(w.keywords)) if isinstance(w, functools.partial) else (w, (), {})
        # recognised by class name AND home (the reference's gym.wrappers.* or this package's): a user class that merely shares a name
        # keeps its own semantics and must reach the error below
        home = (getattr(fn, "__module__", "") or "").split(".")
        name = getattr(fn, "__name__", None) if isinstance(fn, type) and (home[:2] == ["gym", "wrappers"] or home[0] == "gym_amd") else None
        if name == "TimeLimit" and not args:
            k = kw.pop("max_episode_steps", None)
            if kw:
                raise NotImplementedError(f"TimeLimit arguments {sorted(kw)} are not supported by the device engine")
            if k is not None:
                limit = int(k) if limit is None else min(limit, int(k))
        elif name == "RecordEpisodeStatistics" and not args and set(kw) <= {"deque_size"}:
            post.append(("episode_statistics", kw))
        elif name in ("OrderEnforcing", "PassiveEnvChecker") and not args and not kw:
            continue
        elif name in ("ClipAction", "FlattenObservation") and not args and not kw:
            post.append(("identity_for_classic_control", {"wrapper": name}))
        elif name == "RescaleAction" and not args and set(kw) == {"min_action", "max_action"}:
            post.append(("rescale_action", kw))
        elif name in ("TransformObservation", "TransformReward") and ((len(args) == 1 and not kw) or (not args and set(kw) == {"f"})):
            post.append(("transform_observation" if name == "TransformObservation" else "transform_reward", {"f": args[0] if args else kw["f"]}))
        elif name == "NormalizeObservation" and not args and set(kw) <= {"epsilon"}:
            post.append(("normalize_observation", kw))
        elif name == "NormalizeReward" and not args and set(kw) <= {"gamma", "epsilon"}:
            post.append(("normalize_reward", kw))
        else:
            raise NotImplementedError(
                f"per-sub-environment wrapper {w!r} cannot run inside the device engine (no Python sub-envs); recognised: TimeLimit, "
                "RecordEpisodeStatistics, NormalizeObservation, NormalizeReward, ClipAction, RescaleAction, TransformObservation, TransformReward, FlattenObservation, OrderEnforcing, "
                "PassiveEnvChecker as classes or functools.partial — otherwise wrap the vector env "
                "(gym_amd.VectorEnvWrapper) or pass the env's own keyword arguments / max_episode_steps")
    return limit, post


def make(id: str, num_envs: int = 1, asynchronous: bool = False, **kwargs) -> VectorEnv:
    """gym.vector.make (gym/vector/__init__.py:12-73) for the engine's ids.  `asynchronous` is accepted and
    ignored: there are no sub-processes, all sub-envs step in one kernel launch.  `wrappers`: see _sub_env_wrappers."""
    kwargs.pop("disable_env_checker", None)
    limit, post = _sub_env_wrappers(kwargs.pop("wrappers", None))
    from . import toy_text

    if id == "Blackjack-v1":
        cls = toy_text.HipBlackjackVectorEnv
    elif id in toy_text.TOY_TEXT_REGISTRY:  # FrozenLake / Taxi / CliffWalking: the table-driven engine
        cls = toy_text.HipTabularVectorEnv
    else:
        cls = HipVectorEnv
    if limit is not None:
        # the outer TimeLimit(k) over the id's own: whichever is shorter ends the episode (both restart at reset)
        own = kwargs.get("max_episode_steps", _default_limit(id))
        kwargs["max_episode_steps"] = limit if own is None or own <= 0 else min(limit, own)
    env = cls(id, num_envs, **kwargs)
    for what, kw in post:
        if what == "identity_for_classic_control":
            box_actions = type(env.single_action_space).__name__ == "Box"
            if not isinstance(getattr(env, "unwrapped", env), HipVectorEnv) or (kw["wrapper"] == "ClipAction" and not box_actions):
                env.close()
                raise NotImplementedError(f"wrappers={kw['wrapper']} is an identity only for the classic-control ids"
                                          + (" with Box actions (clip_action.py:28 asserts a Box action space)" if kw["wrapper"] == "ClipAction" else ""))
            if kw["wrapper"] == "ClipAction" and id.split("/")[-1].startswith("MountainCarContinuous"):
                from .wrappers import SubEnvClipAction

                env = SubEnvClipAction(env)      # the reward's action penalty sees the clipped action (continuous_mountain_car.py:169)
            continue
        if what in ("transform_observation", "transform_reward"):
            from .wrappers import SubEnvTransformObservation, SubEnvTransformReward

            if what == "transform_observation" and not isinstance(getattr(env, "unwrapped", env), HipVectorEnv):
                env.close()
                raise NotImplementedError("wrappers=TransformObservation is mapped for the classic-control ids (Box observations)")
            env = (SubEnvTransformObservation if what == "transform_observation" else SubEnvTransformReward)(env, **kw)
            continue
        if what == "rescale_action":
            if not isinstance(getattr(env, "unwrapped", env), HipVectorEnv) or type(env.single_action_space).__name__ != "Box":
                env.close()
                raise NotImplementedError("wrappers=RescaleAction needs a Box action space (rescale_action.py:45-47): Pendulum-v1, MountainCarContinuous-v0")
            from .wrappers import SubEnvRescaleAction

            env = SubEnvRescaleAction(env, **kw)
            continue
        if what in ("normalize_observation", "normalize_reward"):
            base = getattr(env, "unwrapped", env)
            if not isinstance(base, HipVectorEnv) and what == "normalize_observation":
                # (around a toy_text sub-env the reference's wrapper returns float64 scalars that SyncVectorEnv then writes into the
                # Discrete space's int64 batch — truncated to integers: nothing a caller could want reproduced)
                env.close()
                raise NotImplementedError("wrappers=NormalizeObservation is mapped for the classic-control ids (Box observations); "
                                          "wrap the toy_text vector env with gym_amd.NormalizeObservation instead")
            from .wrappers import SubEnvNormalizeObservation, SubEnvNormalizeReward

            env = (SubEnvNormalizeObservation if what == "normalize_observation" else SubEnvNormalizeReward)(env, **kw)
            continue
        if what == "episode_statistics":      # every engine carries the fused accumulators (the toy_text ones from round 6 on)
            from .wrappers import SubEnvEpisodeStatistics

            env = SubEnvEpisodeStatistics(env, **kw)
    return env


def _default_limit(id: str):
    """The TimeLimit an id is registered with (gym/envs/__init__.py), or None."""
    from . import registration, toy_text

    if id in toy_text.TOY_TEXT_REGISTRY:
        return toy_text.TOY_TEXT_REGISTRY[id].max_episode_steps
    if id == "Blackjack-v1":
        return None
    return registration.spec(id).max_episode_steps
