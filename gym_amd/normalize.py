"""RunningNormalizer — gym.wrappers.NormalizeObservation / NormalizeReward (gym/wrappers/normalize.py:8-144) on device
tensors.

The reference wraps a vector env and, per step() call, folds the batch of N observations (resp. the N discounted
returns) into a RunningMeanStd and maps the batch with the UPDATED statistics.  A learner that keeps trajectories on the
GPU wants the same transformation applied to the [K, N, ...] tensors a fused rollout produced, K batches at a time and in
the wrappers' order — that is `normalize_obs` / `normalize_rewards` below, backed by the mxv_norm_* kernels
(gym_amd/csrc/mxv_norm.hip).  The batch moments are the only cross-env reduction on the hot path, hence the only place a
sharded vector env needs a collective besides the output all-gather: with world_size > 1 every rank reduces its shard to
per-step column sums ([K][2*dim] fp64), the sums are all-gathered (a few KiB), and every rank runs the identical running
update — so all ranks hold bit-identical statistics, equal to those of an unsharded run for power-of-two shards.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch

__all__ = ["RunningNormalizer", "HipNormBackend", "SubEnvNormalizer"]


class HipNormBackend:
    """Statistics + maps of ONE shard on the HIP device (two mxv_norm objects: observations, returns)."""

    def __init__(self, num_envs: int, obs_dim: int, *, device: int = 0, stream: Optional[torch.cuda.Stream] = None):
        from . import _native

        if not torch.cuda.is_available():
            raise RuntimeError("RunningNormalizer needs a HIP device; gym_amd has no CPU fallback")
        self.torch_device = torch.device("cuda", device)
        self.stream = stream
        sp = stream.cuda_stream if stream is not None else 0
        self.obs = _native.Norm(obs_dim, num_envs, device=device, stream=sp)
        self.rew = _native.Norm(1, num_envs, device=device, stream=sp)

    def obs_sums(self, K, x, sums):
        self.obs.obs_sums(K, x, sums)

    def obs_sums_partials(self, K, partials, sums):
        self.obs.obs_sums_partials(K, partials, partials.shape[-2], sums)

    def reward_sums_partials(self, K, partials, sums):
        self.rew.reward_sums_partials(K, partials, partials.shape[-2], sums)

    def returns_ptr(self) -> int:
        return self.rew.returns_ptr()

    def obs_apply(self, K, x, y, epsilon, all_sums, world, total_rows):
        self.obs.obs_apply(K, x, y, y.dtype == torch.float32, epsilon, all_sums, world, total_rows)

    def reward_sums(self, K, reward, terminated, truncated, gamma, sums):
        self.rew.reward_sums(K, reward, reward.dtype == torch.float32, terminated, truncated, gamma, sums)

    def reward_apply(self, K, reward, out, epsilon, all_sums, world, total_rows):
        self.rew.reward_apply(K, reward, reward.dtype == torch.float32, out, epsilon, all_sums, world, total_rows)

    def obs_state(self):
        return self.obs.get_state()

    def reward_state(self):
        return self.rew.get_state(want_returns=True)

    def set_obs_state(self, mean, var, count):
        self.obs.set_state(mean, var, count)

    def set_reward_state(self, mean, var, count, returns=None):
        self.rew.set_state(mean, var, count, returns)

    def close(self):
        self.obs.close()
        self.rew.close()


class RunningNormalizer:
    """num_envs = rows of THIS shard; total_envs = rows of the logical vector env (the reference's batch size).

    normalize_obs(x)               x float32 [K, N, O] or [N, O]  -> same shape, float64 (reference dtype) or float32
    normalize_rewards(r, te, tr)   r float64/float32 [K, N] or [N] + uint8 flags -> same shape and dtype as r
    obs_rms / return_rms           .mean / .var / .count of the running statistics (host copies, as in the reference)
    """

    def __init__(self, num_envs: int, obs_dim: int, *, device: int = 0, stream=None, gamma: float = 0.99,
                 obs_epsilon: float = 1e-8, reward_epsilon: float = 1e-8, world_size: int = 1,
                 total_envs: Optional[int] = None, group=None, backend=None):
        self.num_envs, self.obs_dim = int(num_envs), int(obs_dim)
        self.gamma, self.obs_epsilon, self.reward_epsilon = float(gamma), float(obs_epsilon), float(reward_epsilon)
        self.world_size = int(world_size)
        self.total_envs = int(total_envs) if total_envs is not None else self.num_envs * self.world_size
        self.group = group
        self.backend = backend if backend is not None else HipNormBackend(num_envs, obs_dim, device=device, stream=stream)
        self.stream = getattr(self.backend, "stream", None)
        self._dev = getattr(self.backend, "torch_device", torch.device("cpu"))

    def _ctx(self):
        from contextlib import nullcontext

        return torch.cuda.stream(self.stream) if self.stream is not None else nullcontext()

    def _all_sums(self, sums: torch.Tensor) -> torch.Tensor:
        """[K][2*dim] sums of this shard -> [world][K][2*dim] in rank order (one small all-gather)."""
        if self.world_size == 1:
            return sums.unsqueeze(0)
        import torch.distributed as dist

        out = torch.empty((self.world_size,) + tuple(sums.shape), dtype=sums.dtype, device=sums.device)
        dist.all_gather_into_tensor(out.view(-1), sums.view(-1), group=self.group)
        return out

    def normalize_obs(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, out_dtype=torch.float64,
                      partials: Optional[torch.Tensor] = None) -> torch.Tensor:
        """partials: the [K, leaves, 2 O] float64 column sums the rollout that wrote `x` left behind (DeviceRollout.trajectory_buffers(
        obs_partials=True)): the pass that would read x once more to form them is skipped."""
        assert x.dtype == torch.float32 and x.is_contiguous()
        K = x.shape[0] if x.dim() == 3 else 1
        assert x.numel() == K * self.num_envs * self.obs_dim, (tuple(x.shape), self.num_envs, self.obs_dim)
        with self._ctx():
            if out is None:
                out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
            assert out.is_contiguous() and out.shape == x.shape and out.dtype in (torch.float64, torch.float32)
            sums = torch.empty((K, 2 * self.obs_dim), dtype=torch.float64, device=x.device)
            if partials is not None:
                assert partials.dtype == torch.float64 and partials.is_contiguous() and partials.dim() == 3
                assert partials.shape[0] >= K and partials.shape[2] == 2 * self.obs_dim, tuple(partials.shape)
                self.backend.obs_sums_partials(K, partials, sums)
            else:
                self.backend.obs_sums(K, x, sums)
            self.backend.obs_apply(K, x, out, self.obs_epsilon, self._all_sums(sums), self.world_size, self.total_envs)
        return out

    def normalize_rewards(self, reward: torch.Tensor, terminated: torch.Tensor, truncated: torch.Tensor,
                          out: Optional[torch.Tensor] = None, partials: Optional[torch.Tensor] = None) -> torch.Tensor:
        """partials: the [K, leaves, 2] float64 sums of the discounted returns the rollout that wrote `reward` left behind (it advanced
        this normaliser's running returns itself: DeviceRollout.fuse_reward_normalizer): the pass over rewards and flags is skipped."""
        assert reward.dtype in (torch.float64, torch.float32) and reward.is_contiguous()
        K = reward.shape[0] if reward.dim() == 2 else 1
        assert reward.numel() == K * self.num_envs
        assert terminated.numel() == reward.numel() and truncated.numel() == reward.numel()
        assert terminated.is_contiguous() and truncated.is_contiguous()
        assert terminated.element_size() == 1 and truncated.element_size() == 1
        with self._ctx():
            if out is None:
                out = torch.empty_like(reward)
            assert out.is_contiguous() and out.shape == reward.shape and out.dtype == reward.dtype
            sums = torch.empty((K, 2), dtype=torch.float64, device=reward.device)
            if partials is not None:
                assert partials.dtype == torch.float64 and partials.is_contiguous() and partials.dim() == 3
                assert partials.shape[0] >= K and partials.shape[2] == 2, tuple(partials.shape)
                self.backend.reward_sums_partials(K, partials, sums)
            else:
                self.backend.reward_sums(K, reward, terminated, truncated, self.gamma, sums)
            self.backend.reward_apply(K, reward, out, self.reward_epsilon, self._all_sums(sums), self.world_size,
                                      self.total_envs)
        return out

    # -- the same two maps fed from raw device addresses (one step, HIP backend): what the NumPy wrappers use to normalise the
    #    outputs of a host step where they still sit in device-visible memory (mxv_staging_view) -----------------------------
    def normalize_obs_at(self, x_ptr: int, out: torch.Tensor) -> torch.Tensor:
        """x: float32 [N, O] at device address x_ptr -> out (float64 or float32 [N, O] device tensor)."""
        assert out.is_contiguous() and out.numel() == self.num_envs * self.obs_dim
        with self._ctx():
            sums = torch.empty((1, 2 * self.obs_dim), dtype=torch.float64, device=out.device)
            self.backend.obs.obs_sums(1, x_ptr, sums)
            self.backend.obs.obs_apply(1, x_ptr, out, out.dtype == torch.float32, self.obs_epsilon, self._all_sums(sums),
                                       self.world_size, self.total_envs)
        return out

    def normalize_rewards_at(self, r_ptr: int, te_ptr: int, tr_ptr: int, out: torch.Tensor) -> torch.Tensor:
        """reward [N] of out's dtype at r_ptr, uint8 flags at te_ptr / tr_ptr -> out ([N] device tensor)."""
        assert out.is_contiguous() and out.numel() == self.num_envs and out.dtype in (torch.float64, torch.float32)
        f32 = out.dtype == torch.float32
        with self._ctx():
            sums = torch.empty((1, 2), dtype=torch.float64, device=out.device)
            self.backend.rew.reward_sums(1, r_ptr, f32, te_ptr, tr_ptr, self.gamma, sums)
            self.backend.rew.reward_apply(1, r_ptr, f32, out, self.reward_epsilon, self._all_sums(sums), self.world_size,
                                          self.total_envs)
        return out

    # -- the wrappers' attributes (normalize.py:64-70,117-125) ---------------------------------------------------------
    @property
    def obs_rms(self):
        mean, var, count = self.backend.obs_state()
        return SimpleNamespace(mean=mean, var=var, count=count)

    @property
    def return_rms(self):
        mean, var, count, _ = self.backend.reward_state()
        return SimpleNamespace(mean=mean[0], var=var[0], count=count)

    @property
    def returns(self):
        return self.backend.reward_state()[3]

    def state_dict(self):
        om, ov, oc = self.backend.obs_state()
        rm, rv, rc, ret = self.backend.reward_state()
        return dict(obs_mean=om, obs_var=ov, obs_count=oc, ret_mean=rm, ret_var=rv, ret_count=rc, returns=ret)

    def load_state_dict(self, sd):
        self.backend.set_obs_state(sd["obs_mean"], sd["obs_var"], sd["obs_count"])
        self.backend.set_reward_state(sd["ret_mean"], sd["ret_var"], sd["ret_count"], sd.get("returns"))

    # -- pickling: the statistics travel, the device objects are rebuilt (process-local: no stream, no process group) ------
    def __getstate__(self):
        if not isinstance(self.backend, HipNormBackend):
            raise TypeError("only the HIP backend pickles")
        return dict(num_envs=self.num_envs, obs_dim=self.obs_dim, gamma=self.gamma, obs_epsilon=self.obs_epsilon,
                    reward_epsilon=self.reward_epsilon, world_size=self.world_size, total_envs=self.total_envs,
                    device=self._dev.index or 0, state=self.state_dict())

    def __setstate__(self, d):
        self.__init__(d["num_envs"], d["obs_dim"], device=d["device"], gamma=d["gamma"], obs_epsilon=d["obs_epsilon"],
                      reward_epsilon=d["reward_epsilon"], world_size=d["world_size"], total_envs=d["total_envs"])
        self.load_state_dict(d["state"])

    def close(self):
        self.backend.close()


class SubEnvNormalizer:
    """The PER-SUB-ENV NormalizeObservation / NormalizeReward of `gym.vector.make(id, n, wrappers=[...])` (gym/vector/__init__.py:56-65
    around gym/wrappers/normalize.py:50-145) on device tensors: every sub-env owns its running statistics and folds ONE row per call
    into them — the terminal observation of an episode that ends and the reset observation that follows are two calls.  Backed by the
    mxv_subnorm_* kernels (one lane per sub-env, statistics in registers across the K steps of a trajectory tensor, no cross-env
    reduction: a sharded vector env needs no collective for it).

    normalize_obs(x, final_obs, terminated, truncated)   x float32 [K, N, O] (or [N, O]) -> same shape float32 (the dtype of the
                                                         reference's batched observations; out_dtype=torch.float64 for the unrounded
                                                         results); `final` = the float64 normalised terminal rows (finished sub-envs only)
    normalize_reset_obs(x)                               reset(): one update per sub-env, nobody has finished
    normalize_rewards(r, terminated, truncated)          float64 / float32 [K, N] (or [N]) -> same shape and dtype
    obs_rms / return_rms / returns                       host copies of every sub-env's statistics ([N, O] / [N])"""

    def __init__(self, num_envs: int, obs_dim: int, *, device: int = 0, stream=None, gamma: float = 0.99, obs_epsilon: float = 1e-8,
                 reward_epsilon: float = 1e-8, backend=None):
        self.num_envs, self.obs_dim = int(num_envs), int(obs_dim)
        self.gamma, self.obs_epsilon, self.reward_epsilon = float(gamma), float(obs_epsilon), float(reward_epsilon)
        self.stream = stream
        if backend is None:
            from . import _native

            if not torch.cuda.is_available():
                raise RuntimeError("SubEnvNormalizer needs a HIP device; gym_amd has no CPU fallback")
            sp = stream.cuda_stream if stream is not None else 0
            backend = SimpleNamespace(obs=_native.SubNorm(obs_dim, num_envs, device=device, stream=sp),
                                      rew=_native.SubNorm(1, num_envs, device=device, stream=sp))
        self.backend = backend

    def _ctx(self):
        from contextlib import nullcontext

        return torch.cuda.stream(self.stream) if self.stream is not None else nullcontext()

    def normalize_obs(self, x, final_obs=None, terminated=None, truncated=None, *, out=None, out_dtype=torch.float32, final_out=None):
        assert x.dtype == torch.float32 and x.is_contiguous()
        K = x.shape[0] if x.dim() == 3 else 1
        assert x.numel() == K * self.num_envs * self.obs_dim, (tuple(x.shape), self.num_envs, self.obs_dim)
        for t_ in (final_obs, terminated, truncated):
            assert t_ is None or t_.is_contiguous()
        if terminated is not None:
            assert truncated is not None and terminated.numel() == K * self.num_envs == truncated.numel()
            assert terminated.element_size() == 1 and truncated.element_size() == 1
        if final_obs is not None:
            assert final_obs.dtype == torch.float32 and final_obs.numel() == x.numel() and terminated is not None
        with self._ctx():
            if out is None:
                out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
            assert out.is_contiguous() and out.shape == x.shape and out.dtype in (torch.float32, torch.float64)
            if final_out is None and final_obs is not None:
                final_out = torch.zeros(x.shape, dtype=torch.float64, device=x.device)
            self.backend.obs.observations(K, x, final_obs, terminated, truncated, out, out.dtype == torch.float32, final_out, self.obs_epsilon)
        return (out, final_out) if final_obs is not None else out

    def normalize_reset_obs(self, x, *, out=None, out_dtype=torch.float32):
        return self.normalize_obs(x, out=out, out_dtype=out_dtype)

    def normalize_rewards(self, reward, terminated, truncated, *, out=None):
        assert reward.dtype in (torch.float64, torch.float32) and reward.is_contiguous()
        K = reward.shape[0] if reward.dim() == 2 else 1
        assert reward.numel() == K * self.num_envs == terminated.numel() == truncated.numel()
        assert terminated.is_contiguous() and truncated.is_contiguous() and terminated.element_size() == 1 == truncated.element_size()
        with self._ctx():
            if out is None:
                out = torch.empty_like(reward)
            assert out.is_contiguous() and out.shape == reward.shape and out.dtype == reward.dtype
            self.backend.rew.rewards(K, reward, reward.dtype == torch.float32, terminated, truncated, out, self.gamma, self.reward_epsilon)
        return out

    @property
    def obs_rms(self):
        mean, var, count, _ = self.backend.obs.get_state()
        return SimpleNamespace(mean=mean, var=var, count=count)

    @property
    def return_rms(self):
        mean, var, count, _ = self.backend.rew.get_state()
        return SimpleNamespace(mean=mean[:, 0], var=var[:, 0], count=count)

    @property
    def returns(self):
        return self.backend.rew.get_state()[3]

    def state_dict(self):
        om, ov, oc, _ = self.backend.obs.get_state()
        rm, rv, rc, ret = self.backend.rew.get_state()
        return dict(obs_mean=om, obs_var=ov, obs_count=oc, ret_mean=rm, ret_var=rv, ret_count=rc, returns=ret)

    def load_state_dict(self, sd):
        self.backend.obs.set_state(sd["obs_mean"], sd["obs_var"], sd["obs_count"])
        self.backend.rew.set_state(sd["ret_mean"], sd["ret_var"], sd["ret_count"], sd.get("returns"))

    def close(self):
        self.backend.obs.close()
        self.backend.rew.close()
