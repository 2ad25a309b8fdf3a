"""Vector-env wrappers of the reference that sit directly on the hot path, backed by the engine.

`RecordEpisodeStatistics` mirrors gym.wrappers.RecordEpisodeStatistics (gym/wrappers/record_episode_statistics.py:40-151)
for a HipVectorEnv: same attributes (`return_queue`, `length_queue`, `episode_count`, `episode_returns`,
`episode_lengths`, `t0`) and the same `infos["episode"]` / `infos["_episode"]` layout (:10-37), but the cumulative
returns and lengths are accumulated inside the step kernel (float32 returns exactly like the reference's np.float32
accumulator; the length is the TimeLimit counter), so no Python loop over the N sub-envs runs per step.
`VectorListInfo` mirrors gym.wrappers.VectorListInfo (gym/wrappers/vector_list_info.py:43-111).
`NormalizeObservation` / `NormalizeReward` mirror gym.wrappers.normalize (gym/wrappers/normalize.py:50-144): same
attributes (`obs_rms`, `return_rms`, `returns`, `gamma`, `epsilon`) and result dtypes (float64), with the batch moments,
the running update and the affine map computed by the mxv_norm_* kernels (gym_amd.normalize.RunningNormalizer).
"""
from __future__ import annotations

import time
from collections import deque
from typing import List

import numpy as np

from .vector_env import HipVectorEnv, LazyInfos, _Pending

__all__ = ["RecordEpisodeStatistics", "VectorListInfo", "NormalizeObservation", "NormalizeReward", "ClipAction", "RescaleAction",
           "TransformObservation", "TransformReward", "SubEnvEpisodeStatistics", "SubEnvClipAction", "SubEnvRescaleAction", "SubEnvTransformObservation", "SubEnvTransformReward",
           "SubEnvNormalizeObservation", "SubEnvNormalizeReward"]


class _VectorWrapper:
    """gym.Wrapper surface for a vector env: attribute access falls through to the wrapped env (gym/core.py:227-243)."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(f"accessing private attribute '{name}' is prohibited")
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)

    def step(self, actions):
        return self.env.step(actions)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def close(self):
        return self.env.close()

    def __repr__(self):
        return f"<{type(self).__name__}{self.env}>"


def _has_fused_statistics(env) -> bool:
    """The engine's vector envs: the classic-control adapter and (round 6) the toy_text ones, whose kernels carry the same float32
    return accumulator and report the TimeLimit counter as the episode length."""
    from .toy_text import HipBlackjackVectorEnv, HipTabularVectorEnv

    return isinstance(env, (HipVectorEnv, HipTabularVectorEnv, HipBlackjackVectorEnv))


class RecordEpisodeStatistics(_VectorWrapper):
    def __init__(self, env: HipVectorEnv, deque_size: int = 100):
        chain, e = [], env
        while isinstance(e, _VectorWrapper):
            chain.append(e)
            e = e.env
        if not _has_fused_statistics(e) or any(isinstance(w, VectorListInfo) for w in chain):
            raise TypeError("gym_amd.wrappers.RecordEpisodeStatistics wraps one of the engine's vector envs — HipVectorEnv, "
                            "HipTabularVectorEnv, HipBlackjackVectorEnv: the statistics are accumulated by its kernels — possibly under "
                            "Normalize* wrappers, and needs dict infos")
        super().__init__(env)
        self.num_envs = env.num_envs
        self.is_vector_env = True
        self.t0 = time.perf_counter()
        self.episode_count = 0
        self.return_queue = deque(maxlen=deque_size)
        self.length_queue = deque(maxlen=deque_size)
        self._enabled = False
        # The engine's fused accumulator sums the RAW rewards of the dynamics.  The reference accumulates whatever the wrapped
        # env's step() returns (record_episode_statistics.py:119-121), so with a NormalizeReward underneath the episode returns
        # are sums of NORMALISED rewards: in that stacking order the returns are accumulated here on the host from the wrapped
        # step's rewards (one vectorised float32 add per step, the reference's own arithmetic); lengths stay the TimeLimit counter.
        self._host_returns = any(isinstance(w, (NormalizeReward, SubEnvNormalizeReward, SubEnvTransformReward)) for w in chain)
        self._acc = None

    # episode_returns / episode_lengths are None before the first reset (record_episode_statistics.py:89-90)
    @property
    def episode_returns(self):
        if not self._enabled:
            return None
        if self._host_returns:
            return self._acc.copy()
        return self.env.handle.episode_stats_host(want_running=True)[2]

    @property
    def episode_lengths(self):
        if not self._enabled:
            return None
        return self.env.handle.get_state()[1].astype(np.int32)

    def reset(self, **kwargs):
        if not self._enabled:
            self.env.handle.episode_stats(True)
            self._enabled = True
        if self._host_returns:
            self._acc = np.zeros(self.num_envs, dtype=np.float32)   # :91-94
        return self.env.reset(**kwargs)  # the engine zeroes the accumulators of every env it resets (:91-94)

    def step(self, action):
        observations, rewards, terminateds, truncateds, infos = self.env.step(action)
        assert isinstance(infos, dict), (f"`info` dtype is {type(infos)} while supported dtype is `dict`. This may be "
                                         "due to usage of other wrappers in the wrong order.")
        base = self.env.unwrapped
        if getattr(base, "_packed", False) and not self._host_returns and isinstance(infos, LazyInfos):
            self._step_packed(base, infos)
            return observations, rewards, terminateds, truncateds, infos
        done = terminateds | truncateds
        if self._host_returns:
            self._acc += rewards   # float32 array += float64 rewards, like the reference's accumulator
        if done.any():
            r, l = self.env.handle.episode_stats_host()
            if self._host_returns:
                r = self._acc.copy()
                self._acc[done] = 0
            t = round(time.perf_counter() - self.t0, 6)
            # add_vector_episode_statistics (:10-37): float64 arrays of length N, zero where no episode ended
            episode = {"r": np.where(done, r, 0).astype(np.float64), "l": np.where(done, l, 0).astype(np.float64),
                       "t": np.where(done, t, 0.0)}
            if isinstance(infos, LazyInfos):
                dict.__setitem__(infos, "episode", episode)
                dict.__setitem__(infos, "_episode", done.copy())
            else:
                infos["episode"], infos["_episode"] = episode, done.copy()
            idx = np.flatnonzero(done)
            self.return_queue.extend(r[idx].tolist())
            self.length_queue.extend(l[idx].tolist())
            self.episode_count += int(idx.size)
        return observations, rewards, terminateds, truncateds, infos

    def _step_packed(self, base, infos):
        """Large vector envs: the device packed (env index, episode return, episode length) of the envs that finished this step
        next to their final observations (mxv_final_packed_stats_view) — a few percent of N.  The queues and the count are
        updated from those; the dense float64 arrays of infos["episode"] (add_vector_episode_statistics, :10-37: length N, zero
        where no episode ended) are built on first access only — three 8-MB arrays per step at 2^20 envs that most loops read
        only through the `_episode` mask."""
        if "_final_observation" not in infos:      # nobody finished
            return
        done = dict.__getitem__(infos, "_final_observation")
        idx, r, l = base.handle.final_packed_stats()
        t = round(time.perf_counter() - self.t0, 6)
        n = self.num_envs

        def build_episode():
            er, el, et = np.zeros(n), np.zeros(n), np.zeros(n)
            er[idx], el[idx], et[idx] = r, l, t
            return {"r": er, "l": el, "t": et}

        dict.__setitem__(infos, "episode", _Pending(build_episode))
        dict.__setitem__(infos, "_episode", _Pending(done.copy))
        keep = self.return_queue.maxlen      # a bounded deque only ever sees the last `maxlen` of them anyway
        self.return_queue.extend((r if keep is None else r[-keep:]).tolist())
        self.length_queue.extend((l if keep is None else l[-keep:]).tolist())
        self.episode_count += int(idx.size)


class SubEnvEpisodeStatistics(_VectorWrapper):
    """What `gym.vector.make(id, n, wrappers=RecordEpisodeStatistics)` yields in the reference (gym/vector/__init__.py:56-65: the wrapper
    around EVERY sub-env): the sub-env whose episode ends reports `{"episode": {"r": float32, "l": int32, "t": float}}` in its own info
    (record_episode_statistics.py:125-136, non-vector branch), and because SyncVectorEnv autoresets that env in the same step, the info
    travels in `infos["final_info"][i]` (sync_vector_env.py:152-156).  Same numbers as the vector-level wrapper — the engine's fused
    float32 accumulators and TimeLimit counters — delivered where the per-sub-env wrapper puts them."""

    def __init__(self, env, deque_size: int = 100):
        super().__init__(RecordEpisodeStatistics(env, deque_size))

    def step(self, action):
        obs, rew, term, trunc, infos = self.env.step(action)
        if "_episode" not in infos:
            return obs, rew, term, trunc, infos
        ep, mask = infos.pop("episode"), infos.pop("_episode")
        idx = np.flatnonzero(mask)
        base = dict.__getitem__(infos, "final_info") if isinstance(infos, LazyInfos) else infos["final_info"]

        def build():
            arr = base.build() if isinstance(base, _Pending) else base
            for i in idx:
                arr[i] = {**(arr[i] or {}), "episode": {"r": np.float32(ep["r"][i]), "l": np.int32(ep["l"][i]), "t": float(ep["t"][i])}}
            return arr

        if isinstance(infos, LazyInfos):
            dict.__setitem__(infos, "final_info", _Pending(build))
        else:
            infos["final_info"] = build()
        return obs, rew, term, trunc, infos


class SubEnvClipAction(_VectorWrapper):
    """`wrappers=ClipAction` (gym/wrappers/clip_action.py:33-43 around every sub-env) where it is NOT an identity:
    MountainCarContinuous-v0 charges `action[0] ** 2 * 0.1` on the action as given (continuous_mountain_car.py:169), so an out-of-range
    action must reach the engine clipped to the action space's bounds."""

    def step(self, actions):
        sp = self.env.single_action_space
        a = np.asarray(actions, dtype=sp.dtype).reshape((self.env.num_envs,) + sp.shape)
        return self.env.step(np.clip(a, sp.low, sp.high))


class SubEnvRescaleAction(_VectorWrapper):
    """`wrappers=partial(RescaleAction, min_action=a, max_action=b)` (gym/wrappers/rescale_action.py:31-82 around every sub-env): the vector
    env's action space becomes Box(a, b) per sub-env, and an action is mapped affinely onto the sub-env's own bounds and clipped to them —
    the reference's float32 expression, operation for operation, on the whole batch; actions outside [a, b] raise its AssertionError."""

    def __init__(self, env, min_action, max_action):
        from .spaces import Box, batch_space

        sp = env.single_action_space
        assert type(sp).__name__ == "Box", f"expected Box action space, got {type(sp)}"                   # :45-47
        assert np.less_equal(min_action, max_action).all(), (min_action, max_action)                       # :48
        super().__init__(env)
        self.num_envs = env.num_envs
        self.is_vector_env = True
        self._low, self._high = sp.low, sp.high
        self.min_action = np.zeros(sp.shape, dtype=sp.dtype) + min_action                                  # :51-56
        self.max_action = np.zeros(sp.shape, dtype=sp.dtype) + max_action
        self.single_action_space = Box(low=min_action, high=max_action, shape=sp.shape, dtype=sp.dtype)    # :57-62
        self.action_space = batch_space(self.single_action_space, self.num_envs)

    def step(self, actions):
        a = np.asarray(actions, dtype=self._low.dtype).reshape((self.num_envs,) + self._low.shape)
        assert np.all(np.greater_equal(a, self.min_action)), (a, self.min_action)                         # :73-76
        assert np.all(np.less_equal(a, self.max_action)), (a, self.max_action)                            # :77
        low, high = self._low, self._high
        a = low + (high - low) * ((a - self.min_action) / (self.max_action - self.min_action))            # :80-82
        return self.env.step(np.clip(a, low, high))                                                        # :83


class _PerEnvMeanStd:
    """N independent RunningMeanStd objects (gym/wrappers/normalize.py:8-48), one per sub-env, each updated with batches of ONE row — what
    `NormalizeObservation(sub_env)` / `NormalizeReward(sub_env)` keep — as arrays over the env axis.  The update is the reference's
    `update_mean_var_count_from_moments` with batch_mean = the row (the float32 / float64 mean of one element is the element), batch_var = 0,
    batch_count = 1, operation for operation and in its order, so every env's statistics are bit-identical to its own wrapper's."""

    def __init__(self, n: int, shape=()):
        self.mean = np.zeros((n,) + tuple(shape), np.float64)
        self.var = np.ones((n,) + tuple(shape), np.float64)
        self.count = np.full(n, 1e-4, np.float64)
        self._bc = (slice(None),) + (None,) * len(shape)

    def update(self, rows, idx=slice(None)):
        mean, var, count = self.mean[idx], self.var[idx], self.count[idx][self._bc]
        delta = rows - mean                                           # :37
        tot = count + 1                                               # :38
        self.mean[idx] = mean + delta * 1 / tot                       # :40
        m2 = var * count + np.float32(0.0) * 1 + np.square(delta) * count * 1 / tot      # :41-43 (m_b = batch_var * batch_count = 0)
        self.var[idx] = m2 / tot                                      # :44
        self.count[idx] = tot[(slice(None),) + (0,) * (tot.ndim - 1)]


class VectorListInfo(_VectorWrapper):
    """Converts the dict-of-arrays infos of a vector env into a list of per-env dicts (vector_list_info.py:43-111)."""

    def __init__(self, env):
        assert getattr(env, "is_vector_env", False), "This wrapper can only be used in vectorized environments."
        super().__init__(env)
        self.num_envs = env.num_envs
        self.is_vector_env = True

    def step(self, action):
        observation, reward, terminated, truncated, infos = self.env.step(action)
        return observation, reward, terminated, truncated, self._convert_info_to_list(infos)

    def reset(self, **kwargs):
        obs, infos = self.env.reset(**kwargs)
        return obs, self._convert_info_to_list(infos)

    def _convert_info_to_list(self, infos: dict) -> List[dict]:
        infos = dict(infos.items())
        list_info = [{} for _ in range(self.num_envs)]
        episode = infos.pop("episode", False)
        if episode:
            mask = infos.pop("_episode")
            for i in np.flatnonzero(mask):
                list_info[i]["episode"] = {k: episode[k][i] for k in ("r", "l", "t")}
        for k in infos:
            if k.startswith("_"):
                continue
            for i in np.flatnonzero(infos[f"_{k}"]):
                list_info[i][k] = infos[k][i]
        return list_info


def _hip_base(env, who: str):
    """The engine's vector env under the wrappers: the classic-control adapter, or (round 6) a toy_text one — the reference's Normalize*
    take any vector env (normalize.py:57-70, :104-121); the statistics live on the env's device either way."""
    base = getattr(env, "unwrapped", env)
    if not _has_fused_statistics(base):
        raise TypeError(f"gym_amd.wrappers.{who} wraps one of the engine's vector envs (the statistics live on its device)")
    return base


class _StagedIO:
    """Host I/O of the Normalize* wrappers.  The step underneath just brought its outputs to the host, but they also still sit
    in device-visible memory (mxv_staging_view): the normaliser reads them THERE instead of uploading the NumPy arrays again,
    and its result comes back with one DMA into a pooled pinned array (views handed to the caller, recycled when dropped —
    gym_amd/_native.py: _BlockPool) instead of a fresh pageable tensor per step.  At 2^20 CartPole envs NormalizeObservation
    went from 5.3 to ~1.6 ms per step that way.  Falls back to uploading the arrays when an inner wrapper altered them."""

    def _staged_setup(self, base, inner_ok):
        e, ok = self.env, True
        while isinstance(e, _VectorWrapper):
            ok = ok and isinstance(e, inner_ok)
            e = e.env
        self._staged = ok and hasattr(base.handle, "staging_view")
        self._base = base
        self._pools = {}

    def _to_host(self, dev_tensor, shape, dtype):
        """Device tensor -> NumPy array of `shape` / `dtype` (same bytes), through a pooled pinned block when one is free."""
        from . import _native

        nbytes = dev_tensor.numel() * dev_tensor.element_size()
        pool = self._pools.get(nbytes)
        if pool is None:
            pool = self._pools[nbytes] = _native.pinned_pool(nbytes)
        raw = pool.take()
        if raw is None:
            return dev_tensor.cpu().numpy().reshape(shape)
        out = raw.view(dtype).reshape(shape)
        self._torch.from_numpy(out).copy_(dev_tensor.view(out.shape))
        return out

    def __getstate__(self):
        d = dict(self.__dict__)
        for k in ("_torch", "_pools", "_out_dev"):
            d.pop(k, None)
        return d

    def __setstate__(self, d):
        import torch

        self.__dict__.update(d)
        self._torch = torch
        self._pools = {}


# Per-sub-env Normalize* (what `make(wrappers=[NormalizeObservation, NormalizeReward])` maps to): from this many sub-envs on, every
# sub-env's running statistics live on the device and are advanced by the mxv_subnorm_* kernels (one lane per sub-env); below it
# they are NumPy arrays on the host — the same arithmetic in the same order, bit for bit (tests/test_gpu_subnorm.py).
SUBENV_DEVICE_MIN = 4096


class _SubEnvDevice(_StagedIO):
    """Device side of SubEnvNormalizeObservation / SubEnvNormalizeReward: one _native.SubNorm, the staged outputs of the host step as
    its inputs (no second upload), pooled pinned arrays for its results."""

    def _device_setup(self, env, dim, inner_ok, device):
        base = getattr(env, "unwrapped", env)
        on = device if device is not None else (_has_fused_statistics(base) and env.num_envs >= SUBENV_DEVICE_MIN)
        self._sub = None
        if not on:
            return False
        import torch

        from . import _native

        if not _has_fused_statistics(base):
            raise TypeError("the device form of the per-sub-env Normalize* wrappers needs one of the engine's vector envs underneath")
        self._torch = torch
        self._dev = torch.device("cuda", base.handle.device)
        self._sub = _native.SubNorm(dim, env.num_envs, device=base.handle.device)
        self._staged_setup(base, inner_ok)
        self._staged = self._staged and hasattr(base.handle, "staging_final")
        return True

    def _dev_buf(self, name, shape, dtype):
        b = self.__dict__.get(name)
        if b is None:
            b = self.__dict__[name] = self._torch.empty(shape, dtype=dtype, device=self._dev)
        return b

    def _up(self, a, dtype):
        return self._torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self._dev)

    def __getstate__(self):
        d = _StagedIO.__getstate__(self)
        sub = d.pop("_sub", None)
        for k in [k for k in d if k.startswith("_buf_")]:
            d.pop(k)
        d["_sub_state"] = None if sub is None else (sub.dim, sub.get_state())
        return d

    def __setstate__(self, d):
        st = d.pop("_sub_state", None)
        _StagedIO.__setstate__(self, d)
        self._sub = None
        if st is not None:
            from . import _native

            base = getattr(self.env, "unwrapped", self.env)
            self._dev = self._torch.device("cuda", base.handle.device)
            self._sub = _native.SubNorm(st[0], self.env.num_envs, device=base.handle.device)
            self._sub.set_state(*st[1])

    def close(self):
        if self._sub is not None:
            self._sub.close()
        return self.env.close()


class SubEnvNormalizeObservation(_SubEnvDevice, _VectorWrapper):
    """What `gym.vector.make(id, n, wrappers=NormalizeObservation)` yields in the reference (gym/vector/__init__.py:56-65 around
    gym/wrappers/normalize.py:50-93): every sub-env normalises its observations with ITS OWN running statistics, updated with one row per
    call — the terminal observation of an episode and the reset observation that follows it are two calls of that env's wrapper
    (step, then the autoreset's reset: sync_vector_env.py:152-156), the batched observations are the float32 cast of the float64 results
    (the vector env's observation space stays float32: numpy_utils.py:49-50 writes into it) and `final_observation` holds the float64
    arrays.  A different normalisation from the vector-level `NormalizeObservation` (batch statistics over all sub-envs).  From
    SUBENV_DEVICE_MIN sub-envs on the statistics live on the device (mxv_subnorm_observations reads the step's outputs where the host
    step left them and one DMA brings the result back); below, host-side NumPy over the arrays the adapter hands back.  Both are the
    reference's arithmetic in its order: bit-identical to each other and to the reference on the same inputs.  `device=True / False`
    forces either."""

    def __init__(self, env, epsilon: float = 1e-8, device=None):
        super().__init__(env)
        self.epsilon = epsilon
        shape = env.single_observation_space.shape
        if not self._device_setup(env, int(shape[0]), (RecordEpisodeStatistics, SubEnvEpisodeStatistics, SubEnvNormalizeReward), device):
            self.obs_rms = _PerEnvMeanStd(env.num_envs, shape)

    def __getattr__(self, name):
        if name == "obs_rms" and self.__dict__.get("_sub") is not None:      # host copies of every sub-env's statistics, like the wrappers' attribute
            from types import SimpleNamespace

            mean, var, count, _ = self._sub.get_state()
            return SimpleNamespace(mean=mean, var=var, count=count)
        return _VectorWrapper.__getattr__(self, name)

    def _normalize(self, rows, idx=slice(None)):
        self.obs_rms.update(rows, idx)
        return (rows - self.obs_rms.mean[idx]) / np.sqrt(self.obs_rms.var[idx] + self.epsilon)       # :90-93

    def reset(self, **kwargs):
        obs, infos = self.env.reset(**kwargs)
        if self._sub is None:
            return self._out(self._normalize(obs), obs), infos
        t = self._torch
        wide = self.__dict__.get("_wide", False)
        y = self._dev_buf("_buf_y", obs.shape, t.float64 if wide else t.float32)
        x = self._base.handle.staging_view()[0] if self._staged else self._up(obs, np.float32)
        self._sub.observations(1, x, None, None, None, y, not wide, None, self.epsilon)
        return self._to_host(y, obs.shape, np.float64 if wide else np.float32), infos

    def step(self, action):
        obs, rew, term, trunc, infos = self.env.step(action)
        done = term | trunc
        if self._sub is not None:
            return self._step_device(obs, rew, term, trunc, infos, done)
        if not done.any():
            return self._out(self._normalize(obs), obs), rew, term, trunc, infos
        idx = np.flatnonzero(done)
        fin = infos["final_observation"]
        first = obs.copy()                                            # what every sub-env's step() returned: terminal rows where it ended
        first[idx] = np.stack([fin[i] for i in idx])
        y = self._normalize(first)
        new_fin = np.full(len(done), None, dtype=object)
        for i in idx:
            new_fin[i] = y[i].copy()                                  # float64, as the sub-env's wrapper returned it
        y[idx] = self._normalize(obs[idx], idx)                       # ... then each finished sub-env's reset(): its second update
        self._set_final(infos, new_fin)
        return self._out(y, obs), rew, term, trunc, infos

    @staticmethod
    def _set_final(infos, new_fin):
        if isinstance(infos, LazyInfos):
            dict.__setitem__(infos, "final_observation", new_fin)
        else:
            infos["final_observation"] = new_fin

    def _out(self, y, obs):
        """The batch as the vector env hands it out: the float32 cast of the float64 results (numpy_utils.py:49-50 writes them into the
        float32 observation buffer) — unless an observation transform sits right above (`_wide`, set by SubEnvTransformObservation): the
        reference's TransformObservation receives the float64 rows and the cast happens after it."""
        return y if self.__dict__.get("_wide", False) else y.astype(obs.dtype)

    def _step_device(self, obs, rew, term, trunc, infos, done):
        t = self._torch
        idx = np.flatnonzero(done)
        wide = self.__dict__.get("_wide", False)
        y = self._dev_buf("_buf_y", obs.shape, t.float64 if wide else t.float32)
        yfin = self._dev_buf("_buf_yfin", obs.shape, t.float64)
        if self._staged:
            h = self._base.handle
            x, _, te, tr = h.staging_view()
            fin = h.staging_final()
        else:                                                         # an inner wrapper altered the arrays: normalise what step() returned
            x, te, tr = self._up(obs, np.float32), self._up(term, np.uint8), self._up(trunc, np.uint8)
            dense = np.zeros(obs.shape, np.float32)
            if idx.size:
                f = infos["final_observation"]
                dense[idx] = np.stack([f[i] for i in idx])
            fin = self._up(dense, np.float32)
        self._sub.observations(1, x, fin, te, tr, y, not wide, yfin, self.epsilon)
        if idx.size:
            rows = yfin[t.from_numpy(idx).to(self._dev)].cpu().numpy()       # the float64 rows of the finished sub-envs only
            n = len(done)

            def build():
                arr = np.full(n, None, dtype=object)
                for j, i in enumerate(idx):
                    arr[i] = rows[j]
                return arr

            if isinstance(infos, LazyInfos):
                dict.__setitem__(infos, "final_observation", _Pending(build))
            else:
                infos["final_observation"] = build()
        return self._to_host(y, obs.shape, np.float64 if wide else np.float32), rew, term, trunc, infos


class _RowMap:
    """A user function the reference applies to ONE sub-env's value (an observation row / a scalar reward), applied to a batch.  The first
    batch decides how: f(batch) is compared with f(row) of (up to 256 of) its rows — equal shape, dtype and bits, as for elementwise maps
    such as `lambda o: np.clip(o, -10, 10)` — and from then on f sees whole batches; any difference or exception keeps the row-by-row
    application, which is what the reference's N Python sub-envs do.  (A function with side effects is called twice on those first rows.)"""

    def __init__(self, f):
        assert callable(f)
        self.f, self.batched = f, None

    def __call__(self, x):
        if self.batched:
            return np.asarray(self.f(x))
        if self.batched is None:
            k = min(len(x), 256)
            rows = [np.asarray(self.f(x[i])) for i in range(k)]
            try:
                whole = np.asarray(self.f(x))
                self.batched = bool(whole.shape[:1] == (len(x),) and all(whole[i].shape == rows[i].shape and whole[i].dtype == rows[i].dtype
                                                                         and np.array_equal(whole[i], rows[i], equal_nan=True) for i in range(k)))
            except Exception:  # noqa: BLE001
                self.batched = False
            if self.batched:
                return whole
            return np.stack(rows + [np.asarray(self.f(x[i])) for i in range(k, len(x))])
        return np.stack([np.asarray(self.f(r)) for r in x])


class SubEnvTransformObservation(_VectorWrapper):
    """`wrappers=partial(TransformObservation, f=...)` (gym/wrappers/transform_observation.py:34-43 around every sub-env): f on every
    sub-env's observation — the terminal one of a finished episode (-> `final_observation`) and the reset one that follows — in the dtype the
    wrapper underneath produces (float64 rows above NormalizeObservation, as in the reference's chain), the batch cast to the observation
    space's dtype afterwards (the space is not changed by the wrapper; numpy_utils.py:49-50).  See _RowMap for how f meets a batch."""

    def __init__(self, env, f):
        super().__init__(env)
        self.f, self._map = f, _RowMap(f)
        self._dtype = env.single_observation_space.dtype
        e = env
        while isinstance(e, (SubEnvNormalizeReward, SubEnvTransformReward, SubEnvEpisodeStatistics, RecordEpisodeStatistics)):      # leave the observations alone
            e = e.env
        if isinstance(e, SubEnvNormalizeObservation):
            e._wide = True

    def reset(self, **kwargs):
        obs, infos = self.env.reset(**kwargs)
        return self._map(obs).astype(self._dtype), infos

    def step(self, action):
        obs, rew, term, trunc, infos = self.env.step(action)
        done = term | trunc
        if done.any() and "final_observation" in infos:
            fin = infos["final_observation"]
            new_fin = np.full(len(done), None, dtype=object)
            for i in np.flatnonzero(done):
                new_fin[i] = self.f(fin[i])
            SubEnvNormalizeObservation._set_final(infos, new_fin)
        return self._map(obs).astype(self._dtype), rew, term, trunc, infos


class SubEnvTransformReward(_VectorWrapper):
    """`wrappers=partial(TransformReward, f=...)` (gym/wrappers/transform_reward.py:36-44 around every sub-env): f on every sub-env's
    reward, the batch a float64 array as SyncVectorEnv keeps it (sync_vector_env.py:66).  See _RowMap."""

    def __init__(self, env, f):
        super().__init__(env)
        self.f, self._map = f, _RowMap(f)

    def step(self, action):
        obs, rew, term, trunc, infos = self.env.step(action)
        return obs, np.asarray(self._map(rew), dtype=np.float64), term, trunc, infos


# The reference's names for the action / transform wrappers, so that a wrappers list can be written without gym installed:
# make(id, n, wrappers=[ClipAction, partial(TransformReward, f=...)]) recognises them by name and home like gym.wrappers' own classes,
# and applied by hand to one of the engine's vector envs they are the per-sub-env forms above.
class ClipAction(SubEnvClipAction):
    pass


class RescaleAction(SubEnvRescaleAction):
    pass


class TransformObservation(SubEnvTransformObservation):
    pass


class TransformReward(SubEnvTransformReward):
    pass


class SubEnvNormalizeReward(_SubEnvDevice, _VectorWrapper):
    """`wrappers=NormalizeReward` (gym/wrappers/normalize.py:96-145 around every sub-env): per-env discounted return, per-env running
    variance of it (batches of one), reward / sqrt(var + epsilon), the return zeroed where the episode ended.  On the device from
    SUBENV_DEVICE_MIN sub-envs on (mxv_subnorm_rewards), host-side NumPy below; exact either way, see SubEnvNormalizeObservation."""

    def __init__(self, env, gamma: float = 0.99, epsilon: float = 1e-8, device=None):
        super().__init__(env)
        self.gamma, self.epsilon = gamma, epsilon
        if not self._device_setup(env, 1, (RecordEpisodeStatistics, SubEnvEpisodeStatistics, SubEnvNormalizeObservation), device):
            self.return_rms = _PerEnvMeanStd(env.num_envs, ())
            self.returns = np.zeros(env.num_envs)

    def __getattr__(self, name):
        if name in ("return_rms", "returns") and self.__dict__.get("_sub") is not None:
            from types import SimpleNamespace

            mean, var, count, ret = self._sub.get_state()
            return ret if name == "returns" else SimpleNamespace(mean=mean[:, 0], var=var[:, 0], count=count)
        return _VectorWrapper.__getattr__(self, name)

    def step(self, action):
        obs, rew, term, trunc, infos = self.env.step(action)
        if self._sub is not None:
            t = self._torch
            out = self._dev_buf("_buf_r", rew.shape, t.float64)
            if self._staged and rew.dtype == np.float64:
                _, r, te, tr = self._base.handle.staging_view()
            else:
                r, te, tr = self._up(rew, np.float64), self._up(term, np.uint8), self._up(trunc, np.uint8)
            self._sub.rewards(1, r, False, te, tr, out, self.gamma, self.epsilon)
            return obs, self._to_host(out, rew.shape, np.float64), term, trunc, infos
        self.returns = self.returns * self.gamma + rew                 # :132
        self.return_rms.update(self.returns)                           # :144
        rew = rew / np.sqrt(self.return_rms.var + self.epsilon)        # :145
        self.returns[term | trunc] = 0.0                               # :134-135
        return obs, rew, term, trunc, infos


class NormalizeObservation(_StagedIO, _VectorWrapper):
    """gym.wrappers.NormalizeObservation for a HipVectorEnv (normalize.py:50-93): every reset()/step() folds the batch of
    N observations into `obs_rms` and returns (obs - mean) / sqrt(var + epsilon) as float64."""

    def __init__(self, env, epsilon: float = 1e-8):
        import torch

        from .normalize import RunningNormalizer

        base = _hip_base(env, "NormalizeObservation")
        super().__init__(env)
        self.num_envs = env.num_envs
        self.is_vector_env = True
        self.epsilon = epsilon
        self._torch = torch
        self._dev = torch.device("cuda", base.handle.device)
        shape = getattr(base.single_observation_space, "shape", None)
        if shape is None:       # Blackjack's Tuple observations: the reference's RunningMeanStd(shape=None) cannot be built either (normalize.py:66-69)
            base_name = type(base.single_observation_space).__name__
            raise TypeError(f"NormalizeObservation needs an observation space with a shape (got {base_name})")
        # Discrete observations (the tabular toy_text ids) are scalars per env: shape () -> one column (exact in float32: < 2^24 states)
        self._rn = RunningNormalizer(self.num_envs, int(shape[0]) if shape else 1, device=base.handle.device, obs_epsilon=epsilon)
        # inner wrappers that leave the observations alone: then the base env's staged observations ARE what step() returned
        self._staged_setup(base, (RecordEpisodeStatistics, NormalizeReward))

    @property
    def obs_rms(self):
        return self._rn.obs_rms

    def step(self, action):
        obs, rews, terminateds, truncateds, infos = self.env.step(action)
        return self._normalize_last_step(obs), rews, terminateds, truncateds, infos

    def reset(self, **kwargs):
        obs, info = self.env.reset(**kwargs)
        return self._normalize_last_step(obs), info

    def _normalize_last_step(self, obs):
        """step()/reset() only: `obs` is what the base env's host call JUST returned, so the same values still sit where the GPU can
        read them (mxv_staging_view) and are normalised there instead of being uploaded again.  Anything else goes through
        normalize(), which works on the array it is given."""
        if not self._staged:
            return self.normalize(obs)
        t = self._torch
        out = self.__dict__.get("_out_dev")
        if out is None:
            out = self._out_dev = t.empty(obs.shape, dtype=t.float64, device=self._dev)
        self._rn.normalize_obs_at(self._base.handle.staging_view()[0], out)
        return self._to_host(out, obs.shape, np.float64)

    def normalize(self, obs):
        """The reference's public method (normalize.py:90-93): folds `obs` — the array passed in, whatever it is — into obs_rms and
        returns it normalised."""
        t = self._torch
        obs = np.asarray(obs)
        x = t.from_numpy(np.ascontiguousarray(obs, dtype=np.float32).reshape(self.num_envs, -1)).to(self._dev)
        return self._rn.normalize_obs(x).cpu().numpy().reshape(obs.shape)

    def close(self):
        self._rn.close()
        return self.env.close()


class NormalizeReward(_StagedIO, _VectorWrapper):
    """gym.wrappers.NormalizeReward for a HipVectorEnv (normalize.py:96-144): discounted returns per env, their running
    variance, rewards / sqrt(var + epsilon); the accumulators of finished envs are zeroed."""

    def __init__(self, env, gamma: float = 0.99, epsilon: float = 1e-8):
        import torch

        from .normalize import RunningNormalizer

        base = _hip_base(env, "NormalizeReward")
        super().__init__(env)
        self.num_envs = env.num_envs
        self.is_vector_env = True
        self.gamma = gamma
        self.epsilon = epsilon
        self._torch = torch
        self._dev = torch.device("cuda", base.handle.device)
        self._rn = RunningNormalizer(self.num_envs, 1, device=base.handle.device, gamma=gamma, reward_epsilon=epsilon)
        self._staged_setup(base, (RecordEpisodeStatistics, NormalizeObservation))   # inner wrappers that leave the rewards alone

    @property
    def return_rms(self):
        return self._rn.return_rms

    @property
    def returns(self):
        return self._rn.returns

    def step(self, action):
        obs, rews, terminateds, truncateds, infos = self.env.step(action)
        t = self._torch
        if self._staged and rews.dtype == np.float64:     # the staged rewards are the handle's reward dtype: float64 only (normalize_rewards_at)
            out = self.__dict__.get("_out_dev")
            if out is None:
                out = self._out_dev = t.empty(rews.shape, dtype=t.float64, device=self._dev)
            _, r_ptr, te_ptr, tr_ptr = self._base.handle.staging_view()
            self._rn.normalize_rewards_at(r_ptr, te_ptr, tr_ptr, out)
            return obs, self._to_host(out, rews.shape, np.float64), terminateds, truncateds, infos
        r = t.from_numpy(np.ascontiguousarray(rews, dtype=np.float64)).to(self._dev)
        te = t.from_numpy(np.ascontiguousarray(terminateds).view(np.uint8)).to(self._dev)
        tr = t.from_numpy(np.ascontiguousarray(truncateds).view(np.uint8)).to(self._dev)
        rews = self._rn.normalize_rewards(r, te, tr).cpu().numpy()
        return obs, rews, terminateds, truncateds, infos

    def close(self):
        self._rn.close()
        return self.env.close()
