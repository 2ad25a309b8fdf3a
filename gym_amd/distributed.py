"""ShardedRollout — one logical vector env partitioned over the GPUs of a node, one process per GPU.

Partition rule: env instances never interact (gym/vector/vector_env.py:13-16), so rank r owns
the contiguous global index range [r*N/G, (r+1)*N/G) and steps it with no data-path collective.  The engine's
Philox streams are indexed by GLOBAL env index, hence a 1/2/4/8-way sharding produces bit-identical
trajectories.  The only exchange is what SyncVectorEnv does with np.stack (gym/vector/utils/numpy_utils.py:50):
concatenating the output tensors.  Per north_star that is one RCCL all-gather of the FINAL observation / reward /
terminated / truncated tensors of a rollout chunk — never per step (inbound xGMI is ~7.5x slower than HBM) — and
it is issued asynchronously on RCCL's stream from a snapshot of the outputs so the next chunk's kernels overlap it.
torch.distributed is used for the process group only (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

Two transports for that gather:
  comm="torch" (default)  one packed torch.distributed.all_gather_into_tensor of a snapshot of the four tensors;
  comm="mxv"              the C ABI's own collective (mxv_comm_init / mxv_allgather_outputs, include/mxv.h): RCCL called
                          directly by libmxv.so on a high-priority side stream, four gathers grouped into one launch, received
                          straight into the full (N_total, ...) tensors (no concatenation afterwards).  torch.distributed is
                          then only the courier of the 128-byte RCCL unique id.  Needs the HIP engine.
"""
from __future__ import annotations

import os
from contextlib import nullcontext
from typing import Callable, Optional

import torch
import torch.distributed as dist


def prefer_high_priority_collectives() -> bool:
    """RCCL's kernels on a high-priority stream: a chunk's all-gather then gets CUs as soon as rollout waves retire instead of queueing
    behind the next chunk's long, chip-filling launch.  Measured with the real RCCL call at world size 1 on a 2^17-env shard: +3.4 % per
    chunk with it, +21 % without (profiles/r3/r3k_chunk_overhead_priority.jsonl).  torch reads TORCH_NCCL_HIGH_PRIORITY when it creates a
    process group, and the variable is PROCESS-WIDE — it changes every NCCL group of the process, a learner's included — so nothing sets
    it behind the caller's back (round 3 did, at import): call this before torch.distributed.init_process_group if you want it (bench.py
    does), or export the variable yourself.  An explicit setting of the user's wins.  Returns whether the variable is now "1".
    (comm="mxv", the C ABI's own communicator, always gathers on its own high-priority stream: mxv_comm_init.)"""
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    return os.environ.get("TORCH_NCCL_HIGH_PRIORITY") == "1"


def partition(total_envs: int, world_size: int, rank: int, align: int = 4):
    """(offset, count) of rank's shard; shards are equal and aligned to the engine's env alignment."""
    if total_envs % world_size != 0:
        raise ValueError(f"num_envs={total_envs} must be divisible by the number of GPUs ({world_size})")
    local = total_envs // world_size
    if local % align != 0:
        raise ValueError(f"per-GPU shard ({local} envs) must be a multiple of {align}")
    return rank * local, local


class GatheredOutputs:
    """Result of one packed all-gather.  `shards(i)` are zero-copy views into the receive buffer (one per rank, in
    rank = global-env-index order); `obs / reward / terminated / truncated` (and tuple unpacking) concatenate them
    into the full (N_total, ...) tensors that np.stack yields in the reference — an HBM-local copy made on first
    use only.  Valid until the owner's next gather_async()."""

    names = ("obs", "reward", "terminated", "truncated")

    def __init__(self, owner: "ShardedRollout"):
        self._o = owner
        self._full = {}

    def shards(self, i: int):
        o = self._o
        if o.comm == "mxv":
            return list(o._mxv_recv[i].chunk(o.world_size, dim=0))
        off, nb, dt, shape = o._layout[i]
        return [o._recv[r, off:off + nb].view(dt).view(shape) for r in range(o.world_size)]

    def __repr__(self):
        return f"GatheredOutputs(world={self._o.world_size}, envs={self._o.total_envs}, comm={self._o.comm!r})"

    def full(self, i: int) -> torch.Tensor:
        if self._o.comm == "mxv":      # received in place: [world][N_local, ...] IS the concatenation
            return self._o._mxv_recv[i]
        if i not in self._full:
            with self._o._stream_ctx():
                self._full[i] = torch.cat(self.shards(i), dim=0)
        return self._full[i]

    obs = property(lambda self: self.full(0))
    reward = property(lambda self: self.full(1))
    terminated = property(lambda self: self.full(2))
    truncated = property(lambda self: self.full(3))

    def __iter__(self):
        return iter(self.full(i) for i in range(4))

    def __len__(self):
        return 4

    def __getitem__(self, i):
        return self.full(range(4)[i])


class ShardedRollout:
    """engine_factory(id, num_envs, env_offset=..., seed=..., action_seed=..., **kw) must return an object with
    .reset(seed), .rollout(K, ...), .rollout_per_step(K, ...), .final_tensors() -> (obs, reward, terminated,
    truncated) of the latest step, .synchronize() and .stream (a torch.cuda.Stream or None).  The default is the
    HIP engine (gym_amd.rollout.DeviceRollout)."""

    def __init__(self, id: str, total_envs: int, *, rank: Optional[int] = None, world_size: Optional[int] = None,
                 device: Optional[int] = None, seed: int = 0, action_seed: int = 0, group=None,
                 engine_factory: Optional[Callable] = None, comm: str = "torch", **engine_kwargs):
        if comm not in ("torch", "mxv"):
            raise ValueError(f"comm must be 'torch' or 'mxv', got {comm!r}")
        self.comm = comm
        self.group = group
        distributed = dist.is_available() and dist.is_initialized()
        self.world_size = world_size if world_size is not None else (dist.get_world_size(group) if distributed else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if distributed else 0)
        self.total_envs = int(total_envs)
        self.env_offset, self.local_envs = partition(self.total_envs, self.world_size, self.rank)
        if engine_factory is None:
            from .rollout import DeviceRollout

            engine_factory = DeviceRollout
            if device is None:
                device = torch.cuda.current_device()
            engine_kwargs["device"] = device
        self.engine = engine_factory(id, self.local_envs, env_offset=self.env_offset, seed=seed,
                                     action_seed=action_seed, **engine_kwargs)
        self._pending = None
        self._recv = None
        self._mxv_recv = None
        self._layout = None
        # Snapshots of the chunk's final tensors, TWO sets used alternately: the fused rollout kernel deposits the last step's
        # outputs into the armed set itself (mxv_set_final_snapshot), the gather reads it while the next chunk — armed with the
        # other set — already runs.  Engines without that hook (the CPU stand-in of the gloo tests) get copies instead.
        self._snap = None
        self._works = [None, None]
        self._cur = 0
        self._cur_written = False     # the most recent rollout deposited its finals into set _cur
        self._cur_gathered = True     # set _cur has been handed to a gather since it was written (=> the next rollout flips sets)
        handle = getattr(self.engine, "handle", None)
        self._in_kernel = handle is not None and hasattr(handle, "set_final_snapshot")
        self._force_collective = False   # measurement hook (tools/chunk_overhead.py): issue the real collective at world size 1 too
        if comm == "mxv":
            if handle is None or not hasattr(handle, "comm_init"):
                raise TypeError("comm='mxv' needs the HIP engine (gym_amd.rollout.DeviceRollout): the collective lives in libmxv.so")
            from . import _native

            ids = [_native.comm_unique_id() if self.rank == 0 else None]
            if self.world_size > 1:
                dist.broadcast_object_list(ids, src=0, group=group)   # torch.distributed carries the 128-byte id, nothing else
            handle.comm_init(self.rank, self.world_size, ids[0])

    # ------------------------------------------------------------------------------------------------
    def _stream_ctx(self):
        s = getattr(self.engine, "stream", None)
        return torch.cuda.stream(s) if s is not None else nullcontext()

    def reset(self, seed: Optional[int] = None):
        return self.engine.reset(seed=seed)

    def rollout(self, K: int, **kw):
        """K vector steps of the local shard (no communication); outputs = the last step."""
        self._arm(K)
        return self.engine.rollout(K, **kw)

    def rollout_per_step(self, K: int, **kw):
        """K vector steps of the local shard into [K, N_local, ...] trajectory tensors (no communication)."""
        self._arm(K)
        return self.engine.rollout_per_step(K, **kw)

    # -- snapshots --------------------------------------------------------------------------------------
    def _snapshots(self):
        """The two snapshot sets (allocated on first use; shapes and dtypes of the engine's final tensors).  torch transport:
        each set is ONE packed buffer (obs | reward | terminated | truncated, sections aligned to 256 B) so that the four
        logical gathers travel as one all-gather; mxv transport: four tensors, gathered as one grouped RCCL launch."""
        if self._snap is None:
            finals = self.engine.final_tensors()
            self._layout, off = [], 0
            for t in finals:
                nbytes = t.numel() * t.element_size()
                self._layout.append((off, nbytes, t.dtype, tuple(t.shape)))
                off = (off + nbytes + 255) // 256 * 256
            self._shard_bytes = off
            with self._stream_ctx():
                dev = finals[0].device
                self._snap = []
                for _ in range(2):
                    if self.comm == "mxv":
                        self._snap.append({"views": [torch.empty_like(t) for t in finals]})
                    else:
                        send = torch.empty(off, dtype=torch.uint8, device=dev)
                        self._snap.append({"send": send, "views": [send[o:o + nb].view(dt).view(shape)
                                                                   for o, nb, dt, shape in self._layout]})
                if self.comm == "mxv":
                    self._mxv_recv = [torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                                      for t in finals]
                else:
                    self._recv = torch.empty((self.world_size, off), dtype=torch.uint8, device=dev)
        return self._snap

    def _wait_set(self, idx: int):
        """Order the engine's stream after the gather that last read snapshot set `idx` (GPU-side; the host does not block)."""
        if self.comm == "mxv":
            # sets alternate per gather: the last gather read set _cur, the one before it the other set
            self.engine.handle.allgather_wait(host_sync=False, age=0 if idx == self._cur else 1)
        elif self._works[idx] is not None:
            with self._stream_ctx():
                self._works[idx].wait()
            self._works[idx] = None

    def _arm(self, K: int = 2):
        """Called before every rollout: pick the snapshot set this rollout deposits its final tensors into."""
        if not self._in_kernel:
            self._cur_written = False
            return
        snap = self._snapshots()
        if self._cur_gathered:                      # the current set is (or was) being read by a gather: use the other one
            nxt = 1 - self._cur
            self._wait_set(nxt)                     # ... once the gather that read IT (two gathers ago) is done
            self._cur = nxt
            self._cur_gathered = False
            self.engine.handle.set_final_snapshot(*snap[nxt]["views"])
        self._cur_written = True

    def gather_async(self):
        """Start the all-gather of the latest chunk's final tensors; returns immediately (the next rollout overlaps it)."""
        snap = self._snapshots()
        if not self._cur_written:                   # no in-kernel snapshot of the latest outputs: copy them
            if self._cur_gathered:
                nxt = 1 - self._cur
                self._wait_set(nxt)
                self._cur = nxt
            else:
                self._wait_set(self._cur)
            with self._stream_ctx():
                for dst, src in zip(snap[self._cur]["views"], self.engine.final_tensors()):
                    dst.copy_(src, non_blocking=True)
        cur = snap[self._cur]
        if self.comm == "mxv":
            self.engine.handle.allgather_outputs(*cur["views"], *self._mxv_recv)
            self._pending = ("mxv", self._cur)
        else:
            with self._stream_ctx():               # the collective is ordered after the engine's stream (where the snapshot was written)
                if self.world_size == 1 and not self._force_collective:
                    self._recv.view(-1).copy_(cur["send"], non_blocking=True)
                    self._works[self._cur] = None
                else:
                    self._works[self._cur] = dist.all_gather_into_tensor(self._recv.view(-1), cur["send"], group=self.group,
                                                                         async_op=True)
            # the set index travels with the pending gather: a rollout issued between gather_async() and wait_gather() — the
            # documented overlap pattern — flips _cur to the other set in _arm()
            self._pending = ("torch", self._cur)
        self._cur_gathered = True
        self._cur_written = False

    def wait_gather(self):
        """GatheredOutputs of the last gather_async (unpacks to the full (N_total, ...) obs / reward / terminated /
        truncated), or None.  Orders the engine's stream after the gather; valid until the next gather_async()."""
        if self._pending is None:
            return None
        kind, idx = self._pending
        if kind == "mxv":
            self.engine.handle.allgather_wait(host_sync=False, age=0)   # age counts gathers, not sets: 0 = the last one issued
        elif self._works[idx] is not None:
            with self._stream_ctx():
                self._works[idx].wait()
            self._works[idx] = None
        self._pending = None
        return GatheredOutputs(self)

    def gather(self):
        self.gather_async()
        return self.wait_gather()

    def synchronize(self):
        self.wait_gather()
        for i in (0, 1):
            if self._works[i] is not None:
                self._works[i].wait()
                self._works[i] = None
        self.engine.synchronize()

    def close(self):
        self.wait_gather()
        close = getattr(self.engine, "close", None)
        if close:
            close()

    def make_normalizer(self, obs_dim: Optional[int] = None, **kw):
        """RunningNormalizer (NormalizeObservation / NormalizeReward) over the LOGICAL vector env: the
        batch of every update is all total_envs rows; the shards exchange their per-step column sums (one small
        all-gather per call) and run the identical running update."""
        from .normalize import RunningNormalizer

        if obs_dim is None:
            obs_dim = self.engine.O
        if "backend" not in kw:
            kw.setdefault("device", self.engine.device.index)
            kw.setdefault("stream", getattr(self.engine, "stream", None))
        return RunningNormalizer(self.local_envs, obs_dim, world_size=self.world_size, total_envs=self.total_envs,
                                 group=self.group, **kw)
