"""MixedRollout — heterogeneous batch {CartPole, Pendulum, Acrobot, MountainCar, ...} (BASELINE.json configs[4]).

The reference has no semantics for mixing env kinds in one vector env (it is explicitly unsupported:
gym/vector/vector_env.py:20-23, and SyncVectorEnv raises on mismatched sub-env spaces,
gym/vector/sync_vector_env.py:220-234).  The only consistent definition — and the one the parity tests use — is the
concatenation of homogeneous segments, each equal to its own `SyncVectorEnv`.

Dispatch: one engine (= one C-ABI handle, one HIP stream) per segment per GPU.  The segments' kernels are
launched back to back on their own streams, so the four grids run concurrently and fill the chip together —
"heterogeneous dispatch" is stream-level concurrency, not a mega-kernel with a per-lane switch (which would
serialise the four code paths inside every wave).  Across GPUs each segment is sharded like ShardedRollout
shards a homogeneous env: rank r owns the r-th contiguous slice of EVERY segment, so all ranks carry the same mix
and the same load; Philox streams use the segment-global env index, hence results do not depend on the number of GPUs.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

from .distributed import ShardedRollout

DEFAULT_MIX = ("CartPole-v1", "Pendulum-v1", "Acrobot-v1", "MountainCar-v0")


class MixedRollout:
    """`total_envs` envs split evenly over `ids` (segment s = global indices [s*total/len(ids), ...))."""

    def __init__(self, total_envs: int, ids: Sequence[str] = DEFAULT_MIX, *, rank: Optional[int] = None,
                 world_size: Optional[int] = None, device: Optional[int] = None, seed: int = 0, action_seed: int = 0,
                 group=None, engine_factory: Optional[Callable] = None, **engine_kwargs):
        ids = list(ids)
        if total_envs % len(ids) != 0:
            raise ValueError(f"num_envs={total_envs} must be divisible by the number of env kinds ({len(ids)})")
        self.ids = ids
        self.total_envs = int(total_envs)
        self.segment_envs = self.total_envs // len(ids)
        # distinct seeds per segment: segment s of the mixed batch is NOT a continuation of segment s-1's streams
        self.segments: Dict[str, ShardedRollout] = {}
        for s, env_id in enumerate(ids):
            if env_id in self.segments:
                raise ValueError(f"env id {env_id!r} listed twice")
            self.segments[env_id] = ShardedRollout(env_id, self.segment_envs, rank=rank, world_size=world_size,
                                                   device=device, seed=seed + 1000003 * s,
                                                   action_seed=action_seed + 1000003 * s, group=group,
                                                   engine_factory=engine_factory, **engine_kwargs)
        first = next(iter(self.segments.values()))
        self.rank, self.world_size = first.rank, first.world_size
        self.local_envs = sum(sr.local_envs for sr in self.segments.values())

    def reset(self, seed: Optional[int] = None):
        return {k: sr.reset(seed=None if seed is None else seed + 1000003 * s)
                for s, (k, sr) in enumerate(self.segments.items())}

    def rollout(self, K: int, **kw):
        """K vector steps of every segment; launches are asynchronous and overlap across the segments' streams."""
        return {k: sr.rollout(K, **kw) for k, sr in self.segments.items()}

    def rollout_per_step(self, K: int, out: Optional[dict] = None, **kw):
        return {k: sr.rollout_per_step(K, out=None if out is None else out[k], **kw) for k, sr in self.segments.items()}

    def trajectory_buffers(self, K: int):
        return {k: sr.engine.trajectory_buffers(K) for k, sr in self.segments.items()}

    def gather_async(self):
        for sr in self.segments.values():
            sr.gather_async()

    def wait_gather(self):
        return {k: sr.wait_gather() for k, sr in self.segments.items()}

    def gather(self):
        self.gather_async()
        return self.wait_gather()

    def synchronize(self):
        for sr in self.segments.values():
            sr.synchronize()

    def close(self):
        for sr in self.segments.values():
            sr.close()
