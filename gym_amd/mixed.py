"""MixedRollout — heterogeneous batch {CartPole, Pendulum, Acrobot, MountainCar, ...} (BASELINE.json configs[4]).

The reference has no semantics for mixing env kinds in one vector env (it is explicitly unsupported:
gym/vector/vector_env.py:20-23, and SyncVectorEnv raises on mismatched sub-env spaces,
gym/vector/sync_vector_env.py:220-234).  The only consistent definition — and the one the parity tests use — is the
concatenation of homogeneous segments, each equal to its own `SyncVectorEnv`.

Dispatch: one engine (= one C-ABI handle, one HIP stream) per segment per GPU.  Two ways to run a chunk of K steps, bit-identical:
  * `single_launch=False` (default): the segments' fused rollouts are launched back to back on their own streams; the four
    grids run concurrently.  2.62 us per mixed step at configs[4]'s per-GPU share (4 x 2^15 envs) = 5.0e10 env-steps/s — 94 % of
    the floor set by the Acrobot segment alone (2.47 us: 512 latency-bound RK4 waves, one per SIMD).
  * `single_launch=True`: ONE kernel for all segments (`mxv_rollout_mixed`, include/mxv.h): a block -> segment table sends every
    workgroup (= one wave) to the rollout body of its segment's env kind, so waves stay homogeneous (no per-lane switch that
    would serialise the code paths).  Measured SLOWER on the MI355X: 3.27 us with contiguous block ranges, 4.52 us with the
    ranges interleaved (profiles/r2/r02d_mixed_dispatch_contiguous.jsonl, r02e_mixed_dispatch_interleaved.jsonl): inside one grid the
    hardware deals consecutive workgroups over the SIMDs, which pairs waves of the same segment — two VALU-bound Acrobot waves —
    on one SIMD, while separate grids interleave kinds.  Kept as an option (one launch instead of four matters when launches
    are the bottleneck: short chunks, many small segments).
Across GPUs each segment is sharded like ShardedRollout
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
                 group=None, engine_factory: Optional[Callable] = None, single_launch: bool = False, **engine_kwargs):
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
        # one launch for all segments needs the HIP engine's handles
        self.single_launch = bool(single_launch) and engine_factory is None and len(ids) <= 8
        self._traj = {}

    def _launch_all(self, K: int, outs: dict, per_step: bool) -> bool:
        """All segments in one kernel launch; False if a segment needs its own kernel (the caller falls back)."""
        from . import _native

        engines = [sr.engine for sr in self.segments.values()]
        for sr in self.segments.values():
            sr._arm(K)                      # the segments' final-tensor snapshots for the gather (ShardedRollout)
        for e in engines:
            e._attach_episode_outputs(None)
        try:
            _native.rollout_mixed([e.handle for e in engines], K, [outs[k] for k in self.segments], per_step=per_step)
        except _native.MxvError as err:
            if err.code == _native.ERR_UNSUPPORTED:
                return False
            raise
        return True

    def reset(self, seed: Optional[int] = None):
        return {k: sr.reset(seed=None if seed is None else seed + 1000003 * s)
                for s, (k, sr) in enumerate(self.segments.items())}

    def rollout(self, K: int, **kw):
        """K vector steps of every segment; the output tensors hold the last step (the "final tensors" of the chunk)."""
        if self.single_launch and K > 1 and not kw:
            outs = {k: dict(obs=sr.engine.obs, reward=sr.engine.reward, terminated=sr.engine.terminated,
                            truncated=sr.engine.truncated) for k, sr in self.segments.items()}
            if self._launch_all(K, outs, per_step=False):
                for sr in self.segments.values():
                    e = sr.engine
                    e._last = (e.obs, e.reward, e.terminated, e.truncated)
                return {k: sr.engine._last for k, sr in self.segments.items()}
        return {k: sr.rollout(K, **kw) for k, sr in self.segments.items()}

    def rollout_per_step(self, K: int, out: Optional[dict] = None, **kw):
        """K vector steps of every segment into per-segment [K, N_local, ...] trajectory tensors."""
        if self.single_launch and K > 1 and kw.get("mode", "fused") == "fused" and kw.get("record_actions", True):
            if out is None:
                out = self._traj.get(K) or self._traj.setdefault(K, self.trajectory_buffers(K))
            if all("ep_return" not in o for o in out.values()) and self._launch_all(K, out, per_step=True):
                for k, sr in self.segments.items():
                    o = out[k]
                    sr.engine._last = (o["obs"][K - 1], o["reward"][K - 1], o["terminated"][K - 1], o["truncated"][K - 1])
                return out
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
