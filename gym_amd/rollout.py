"""DeviceRollout — device-resident front-end of the engine (no host round trip per step).

The NumPy contract of HipVectorEnv moves 22 MB per step over PCIe at 2^20 CartPole envs; an RL
learner that lives on the same GPU wants the step outputs where they are.  This class owns torch
tensors for the step I/O (torch is used for device memory and streams only), hands their device
pointers to the C ABI and keeps everything on one HIP stream.  Semantics are those of
SyncVectorEnv.step_wait (gym/vector/sync_vector_env.py:135-169); dtypes are the engine's:
obs float32 (N, O), reward float64 (float32 with reward_f32), terminated/truncated uint8 (N,).

Stream ordering.  Every launch goes to `self.stream`, a stream of its own (so rollouts overlap with a learner's kernels and
with RCCL).  Outputs are therefore ready *on that stream*: consume them inside `with torch.cuda.stream(r.stream):`, after
`r.ready()` (the caller's current stream waits for the engine on the GPU, no host synchronisation) or after
`r.synchronize()` (host wait).  Inputs go the other way: `step(actions)` / `rollout_tape(actions)` first make the engine's
stream wait for the caller's current stream, so actions a policy just computed there are complete before the kernel reads them.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _native
from .registration import spec as _spec


MODES = {"eager": _native.ROLLOUT_EAGER, "graph": _native.ROLLOUT_GRAPH, "fused": _native.ROLLOUT_FUSED}


class DeviceRollout:
    def __init__(self, id: str, num_envs: int, *, device: int = 0, env_offset: int = 0, seed: int = 0,
                 action_seed: int = 0, max_episode_steps: Optional[int] = None, reward_f32: bool = False,
                 action_i32: bool = False, autoreset: bool = True, stream: Optional["torch.cuda.Stream"] = None,
                 obs_carries_state: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceRollout needs a HIP device (torch.cuda.is_available() is False); "
                               "gym_amd has no CPU fallback")
        self.spec = _spec(id)
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", device)
        limit = self.spec.max_episode_steps if max_episode_steps is None else max_episode_steps
        flags = (_native.FLAG_REWARD_F32 if reward_f32 else 0) | (_native.FLAG_ACTION_I32 if action_i32 else 0)
        if not autoreset:
            flags |= _native.FLAG_NO_AUTORESET
        self.handle = _native.Handle(self.spec.kind, num_envs, -1 if limit is None else int(limit), device=device,
                                     env_offset=env_offset, seed=seed, action_seed=action_seed, flags=flags)
        self.O, self.S, self.NA = self.handle.O, self.handle.S, self.handle.NA
        # one torch-visible stream carries every launch of this handle: a stream of its own by default (rollouts then overlap the
        # learner's kernels and RCCL), or the caller's (`stream=`): a learner that steps with its own actions every iteration saves the
        # cross-stream wait of step() that way — ~6 us of GPU-side dependency latency per step, 18.8 instead of 25 us per 2^20-env
        # CartPole step (bench.py variants.step_loop; `with torch.cuda.stream(r.stream):` around the loop does the same)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self.handle.set_stream(self.stream.cuda_stream)
        self.reward_dtype = torch.float32 if reward_f32 else torch.float64
        if self.NA > 0:
            self.action_dtype = torch.int32 if action_i32 else torch.int64
        else:
            self.action_dtype = torch.float32
        with torch.cuda.stream(self.stream):
            n = self.num_envs
            self.obs = torch.empty((n, self.O), dtype=torch.float32, device=self.device)
            self.reward = torch.empty(n, dtype=self.reward_dtype, device=self.device)
            self.terminated = torch.empty(n, dtype=torch.uint8, device=self.device)
            self.truncated = torch.empty(n, dtype=torch.uint8, device=self.device)
            self.final_obs = torch.zeros((n, self.O), dtype=torch.float32, device=self.device)
            self.actions = torch.zeros(n, dtype=self.action_dtype, device=self.device)
        self.stream.synchronize()
        # obs_carries_state=True (CartPole, MountainCar, MountainCarContinuous): `self.obs` doubles as the float32 half of the fp64 state
        # between step() calls (mxv_adopt_obs: float32 observation + int32 residual = the double, exactly) — a quarter less state traffic
        # per step (2^20 CartPole envs: 108 -> 92 bytes per env-step).  The price is a contract: READ self.obs, never write it
        # (in-place normalisation, clamping ... would change the state); every other call hands the state back to fp64 by itself.
        self.obs_carries_state = bool(obs_carries_state)
        if self.obs_carries_state:
            self.handle.adopt_obs(self.obs)
        self._last = (self.obs, self.reward, self.terminated, self.truncated)
        # step(actions) is the learner-in-the-loop call: one launch per policy step, so its host cost is what caps small and
        # medium vector envs.  Everything it needs per call is looked up once here (17.4 -> ~8 us, profiles/HISTORY.md).
        self._stream_ptr = int(self.stream.cuda_stream)
        self._dev_index = self.device.index
        self._step_ptrs = (self.obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(), self.truncated.data_ptr(),
                           self.final_obs.data_ptr())
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)

    def _order_after_caller(self):
        """The engine's stream waits (on the GPU) for what the caller queued on ITS current stream — where the policy wrote the
        actions.  One native call (mxv_wait_stream: event record + stream wait); nothing at all when the caller already
        works on the engine's stream."""
        cur = self._raw_stream(self._dev_index) if self._raw_stream is not None \
            else int(torch.cuda.current_stream(self.device).cuda_stream)
        if cur != self._stream_ptr:
            self.handle.wait_stream(cur)

    # -- reference-shaped calls ------------------------------------------------------------------
    def seed(self, seed: int, action_seed: Optional[int] = None):
        self.handle.seed(seed)
        if action_seed is not None:
            self.handle.seed_actions(action_seed)

    def reset(self, seed: Optional[int] = None, mask: Optional[torch.Tensor] = None, bounds=None) -> torch.Tensor:
        if seed is not None:
            self.handle.seed(seed)
        self.handle.reset(self.obs, mask_dev=mask, bounds=bounds)
        self.ready()        # resets are rare: the returned tensor is safe to read on the caller's current stream (GPU-side ordering, no host wait)
        return self.obs

    def step(self, actions: torch.Tensor, want_final: bool = True):
        """One vector step with caller-provided actions (device tensor of the engine's action dtype)."""
        assert actions.is_cuda and actions.dtype == self.action_dtype and actions.numel() == self.num_envs
        assert actions.is_contiguous()
        self._order_after_caller()                                        # the actions were produced on the caller's stream
        if getattr(self, "episode_stats", False):
            self._attach_episode_outputs(None)
        self._attach_partials(None, None)
        p = self._step_ptrs
        self.handle.step(actions.data_ptr(), p[0], p[1], p[2], p[3], p[4] if want_final else None)
        self._last = (self.obs, self.reward, self.terminated, self.truncated)
        return self._last

    def enable_graph_capture(self, on: bool = True):
        """Make this engine's calls recordable into a hipGraph of the CALLER's (torch.cuda.graph, hipStreamBeginCapture): the vector-step
        index moves into device memory and advances on the stream (mxv_set_device_clock), so that a replayed graph continues the
        action / noise streams where single calls would instead of repeating its capture-time step.  Typical use — a policy in the
        loop at a batch size where launches, not kernels, bound the loop:

            r.enable_graph_capture()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(r.stream):
                step_once()                                   # warm-up outside the capture (allocations, lazy initialisation)
                with torch.cuda.graph(g, stream=r.stream):
                    for _ in range(32):
                        r.step(policy(r.obs))                 # r.obs / r.reward / ... are the engine's own (static) tensors
            for _ in range(1000):
                g.replay()                                    # 32 000 vector steps, 1 000 host calls

        synchronize() / get_counters() still work (outside captures); turning it off reads the index back."""
        self.handle.set_device_clock(on)

    def graphed_loop(self, policy, steps_per_graph: int = 32, *, warmup: int = 3, on_step=None):
        """`steps_per_graph` iterations of  actions = policy(self.obs); self.step(actions)  recorded once into a hipGraph and returned
        as a torch.cuda.CUDAGraph: every `.replay()` advances the vector env by that many steps with one host call.  `policy` maps the
        engine's observation tensor [N, O] to an action tensor of the engine's dtype/shape using torch ops only (no host
        synchronisation, no data-dependent Python control flow: the rules of torch.cuda.graph); `on_step(k)` (optional) runs after
        step k inside the recording — e.g. to copy self.obs / self.reward / ... into the k-th row of the caller's own static
        trajectory tensors.  Everything is recorded on the engine's stream; call self.ready() (or synchronize()) before reading
        results on another stream."""
        self.enable_graph_capture()

        def one(k):
            a = policy(self.obs)
            if a.dtype != self.action_dtype:
                a = a.to(self.action_dtype)
            self.step(a.contiguous(), want_final=False)
            if on_step is not None:
                on_step(k)

        with torch.cuda.stream(self.stream):
            for k in range(warmup):                          # lazy initialisation and the caching allocator's first blocks stay outside
                one(k % steps_per_graph)
            self.stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                for k in range(steps_per_graph):
                    one(k)
        return g

    def step_sampled(self, want_final: bool = False, record_actions: bool = True):
        """One vector step with actions drawn on device (action_space.sample())."""
        self._attach_episode_outputs(None)
        self._attach_partials(None, None)
        self.handle.step_sampled(self.obs, self.reward, self.terminated, self.truncated,
                                 self.final_obs if want_final else None, self.actions if record_actions else None)
        self._last = (self.obs, self.reward, self.terminated, self.truncated)
        return self._last

    def rollout(self, K: int, *, mode: str = "fused", record_actions: bool = False, want_final: bool = False):
        """K sampled steps back to back; the output tensors hold the last step ("final tensors" of the chunk).
        mode: "fused" (one launch, state in registers), "graph" (K launches from a hipGraph) or "eager"."""
        self._attach_episode_outputs(None)
        self._attach_partials(None, None)
        self.handle.rollout(K, self.obs, self.reward, self.terminated, self.truncated,
                            self.final_obs if want_final else None, self.actions if record_actions else None,
                            per_step=False, mode=MODES[mode])
        self._last = (self.obs, self.reward, self.terminated, self.truncated)
        return self._last

    def enable_episode_stats(self):
        """RecordEpisodeStatistics fused into the step kernels: `ep_return` (float32) / `ep_length` (int32) hold, where
        terminated | truncated of a step is set, the return and length of the episode that ended there."""
        self.handle.episode_stats(True)
        self.episode_stats = True
        with torch.cuda.stream(self.stream):
            self.ep_return = torch.zeros(self.num_envs, dtype=torch.float32, device=self.device)
            self.ep_length = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self.stream.synchronize()
        self.handle.set_episode_outputs(self.ep_return, self.ep_length)

    def _attach_episode_outputs(self, out):
        if getattr(self, "episode_stats", False):
            tgt = (out["ep_return"], out["ep_length"]) if out is not None else (self.ep_return, self.ep_length)
            if getattr(self, "_ep_attached", None) is not tgt[0]:
                self.handle.set_episode_outputs(*tgt)
                self._ep_attached = tgt[0]

    def trajectory_buffers(self, K: int, want_final: bool = False, layout: str = "auto",
                           max_park_bytes: Optional[int] = None, obs_partials: bool = False, ret_partials: bool = False):
        """[K, N, ...] output tensors for rollout_per_step (allocate once, reuse every chunk).  want_final adds
        `final_obs` [K, N, O]: info["final_observation"] of every step — rows are written only where terminated | truncated
        of that step is set (what a learner bootstraps from when an episode was truncated), other rows keep their content.

        layout="sorted" (what "auto" picks for sets of 1 GiB and more): ordinary allocations, but the reward / action tensors are
        made to lie in another third of the HBM address space than the observations (_sorted_buffers): the write-bound rollout then
        runs in its fast mode by construction instead of one time in three; the report is left in
        `self.last_placement`.  The search holds extra device memory while it runs (typically a few GiB for 0.1 s): at most
        `max_park_bytes` (default: half of what is free beyond the set and at most 8 GiB whenever anybody else holds device memory or this
        process is one of several ranks; on an otherwise EMPTY device of a single process — MXV_PLACEMENT's default "auto" resolves to
        "search" there — up to 112 GiB for ~3 s, released before the call returns; gym_amd/placement.py has the brakes), it never raises
        on its own account (out of memory inside the search -> ordinary allocations), and MXV_PLACEMENT=off makes "auto" mean
        "separate" for the whole process (gym_amd/placement.py).  layout="placed": the same goal through HIP's virtual-memory API (mxv_placed_alloc, include/mxv.h:
        256-MiB physical chunks of measured class mapped under the tensors) — less transient memory when the classes are interleaved,
        but the real kernel runs 4-10 % slower on memory mapped that way.  layout="separate": one torch allocation per tensor.  (Rounds 1-2's "spread" layout and timing-based
        `tuned_trajectory_buffers` were removed in round 5: sorting by measured HBM class replaced both; profiles/HISTORY.md.)"""
        if obs_partials or ret_partials:
            # + "obs_partials" [K, leaves, 2 O] float64: rollout_per_step then also leaves every step's column sums / sums of squares
            # of the observations per tile of envs (mxv_set_obs_partials) — RunningNormalizer.normalize_obs(x, partials=...) folds them
            # instead of reading the observations a second time;  + "ret_partials" [K, leaves, 2]: the same for NormalizeReward's
            # discounted returns, which the rollout then advances itself (after fuse_reward_normalizer(normalizer))
            out = self.trajectory_buffers(K, want_final, layout, max_park_bytes)
            leaves, _, vals = self.handle.obs_partials_layout()
            with torch.cuda.stream(self.stream):
                if obs_partials:
                    out["obs_partials"] = torch.empty((K, leaves, vals), dtype=torch.float64, device=self.device)
                if ret_partials:
                    out["ret_partials"] = torch.empty((K, leaves, 2), dtype=torch.float64, device=self.device)
            return out
        n, dev = self.num_envs, self.device
        specs = []
        if want_final:
            specs.append(("final_obs", (K, n, self.O), torch.float32, True))
        if getattr(self, "episode_stats", False):
            specs += [("ep_return", (K, n), torch.float32, True), ("ep_length", (K, n), torch.int32, True)]
        specs += [("obs", (K, n, self.O), torch.float32, False), ("reward", (K, n), self.reward_dtype, False),
                  ("terminated", (K, n), torch.uint8, False), ("truncated", (K, n), torch.uint8, False),
                  ("actions", (K, n), self.action_dtype, False)]
        if layout == "auto":
            total = sum(math.prod(shape) * torch.empty((), dtype=dt).element_size() for _, shape, dt, _ in specs)
            from . import placement

            layout = "sorted" if total >= _native.SORTED_MIN_BYTES and placement.enabled() else "separate"   # MXV_PLACEMENT=off: never sort
        if layout == "sorted":
            return self._sorted_buffers(specs, max_park_bytes)
        if layout == "placed":
            npdt = {torch.float32: "<f4", torch.float64: "<f8", torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1"}
            group = {"obs": 0, "reward": 1, "actions": 1}
            mem = _native.PlacedMemory(dev.index, [(name, shape, npdt[dt], group.get(name, -1)) for name, shape, dt, _ in specs])
            self.last_placement = mem.info
            out = mem.tensors()
            with torch.cuda.stream(self.stream):
                for name, _, _, zero in specs:
                    if zero:
                        out[name].zero_()
            return out
        with torch.cuda.stream(self.stream):
            if layout == "separate":
                return {name: (torch.zeros if zero else torch.empty)(shape, dtype=dt, device=dev) for name, shape, dt, zero in specs}
            raise ValueError(f"layout must be 'auto', 'sorted', 'placed' or 'separate', got {layout!r}")

    def _sorted_buffers(self, specs, budget_bytes: Optional[int] = None):
        """Ordinary (torch / hipMalloc) tensors, SORTED by HBM class (gym_amd/placement.py): observations on one class, rewards +
        actions — the same bytes per env-step for CartPole — on another; the report is left in self.last_placement."""
        from .placement import sorted_tensors

        out, report = sorted_tensors(specs, {"obs": 0, "reward": 1, "actions": 1}, self.device, self.stream, budget_bytes)
        self.last_placement = report
        return out

    def state_dict(self) -> dict:
        """Snapshot (NumPy arrays and ints, picklable) from which load_state_dict() continues bit-identically: env state,
        TimeLimit counters, RNG seeds + counters, physics parameters, running episode returns.  Output tensors are not part
        of it (they are rewritten by the next call)."""
        self.stream.synchronize()
        return self.handle.snapshot()

    def load_state_dict(self, snap: dict):
        self.stream.synchronize()
        if snap["stats_on"] and not getattr(self, "episode_stats", False):
            self.enable_episode_stats()
        self.handle.restore(snap)

    def rollout_per_step(self, K: int, *, mode: str = "fused", out: Optional[dict] = None, record_actions: bool = True):
        """K sampled steps, every step's outputs kept in [K, N, ...] trajectory tensors (returned as a dict)."""
        out = self.trajectory_buffers(K) if out is None else out
        assert out["obs"].shape[0] >= K
        self._attach_episode_outputs(out)
        self._attach_partials(out.get("obs_partials"), out.get("ret_partials"))
        self.handle.rollout(K, out["obs"], out["reward"], out["terminated"], out["truncated"], out.get("final_obs"),
                            out["actions"] if record_actions else None, per_step=True, mode=MODES[mode])
        self._last = (out["obs"][K - 1], out["reward"][K - 1], out["terminated"][K - 1], out["truncated"][K - 1])
        return out

    def _attach_partials(self, part, rpart):
        """Attach / detach the buffers of the fused batch moments (mxv_set_obs_partials, mxv_set_return_partials) when they change."""
        if part is not getattr(self, "_partials_attached", None):
            self.handle.set_obs_partials(part)
            self._partials_attached = part
        if rpart is not getattr(self, "_ret_partials_attached", None):
            if rpart is not None and getattr(self, "_fused_returns", None) is None:
                raise RuntimeError("'ret_partials' needs fuse_reward_normalizer(normalizer) first: the rollout advances THAT normaliser's returns")
            ptr, gamma = self._fused_returns if rpart is not None else (None, 0.0)
            self.handle.set_return_partials(ptr, gamma, rpart)
            self._ret_partials_attached = rpart

    def rollout_tape(self, actions: torch.Tensor, *, out: Optional[dict] = None):
        """One fused launch driven by an action tape actions[K, N] (the engine's action dtype)."""
        K = actions.shape[0]
        assert actions.is_cuda and actions.is_contiguous() and actions.dtype == self.action_dtype
        assert actions.numel() == K * self.num_envs
        out = self.trajectory_buffers(K) if out is None else out
        if "obs_partials" in out or "ret_partials" in out:
            raise ValueError("tape-driven launches do not form the batch moments (obs_partials / ret_partials): use buffers without them")
        self._attach_partials(None, None)
        self._order_after_caller()                                        # the tape was produced on the caller's stream
        self._attach_episode_outputs(out)
        self.handle.rollout_tape(K, actions, out["obs"], out["reward"], out["terminated"], out["truncated"],
                                 out.get("final_obs"), per_step=True)
        self._last = (out["obs"][K - 1], out["reward"][K - 1], out["terminated"][K - 1], out["truncated"][K - 1])
        return out

    def fuse_reward_normalizer(self, normalizer):
        """Let trajectory rollouts whose buffers carry "ret_partials" advance `normalizer`'s running discounted returns
        (NormalizeReward, normalize.py:132-136) and leave their per-step sums behind (mxv_set_return_partials); feed them back with
        normalizer.normalize_rewards(reward, terminated, truncated, partials=out["ret_partials"])."""
        self._fused_normalizer = normalizer                      # keeps the returns array the kernels write alive
        self._fused_returns = (normalizer.backend.returns_ptr(), float(normalizer.gamma))
        if getattr(self, "_ret_partials_attached", None) is not None:
            self.handle.set_return_partials(None, 0.0, None)     # re-attached with the new array by the next rollout_per_step
        self._ret_partials_attached = None

    def make_normalizer(self, **kw):
        """RunningNormalizer (NormalizeObservation / NormalizeReward on device tensors) sized for this
        engine and ordered on its stream: feed it the reset observations and the [K, N, ...] trajectory tensors."""
        from .normalize import RunningNormalizer

        return RunningNormalizer(self.num_envs, self.O, device=self.device.index, stream=self.stream, **kw)

    def make_subenv_normalizer(self, **kw):
        """SubEnvNormalizer: the PER-SUB-ENV NormalizeObservation / NormalizeReward of `make(wrappers=[...])` (every sub-env its own running
        statistics, batches of one) on this engine's device tensors and stream; feed it step() outputs or [K, N, ...] trajectory tensors
        (with their "final_obs")."""
        from .normalize import SubEnvNormalizer

        return SubEnvNormalizer(self.num_envs, self.O, device=self.device.index, stream=self.stream, **kw)

    def final_tensors(self):
        """(obs, reward, terminated, truncated) of the most recent vector step (views, valid until the next call)."""
        return self._last

    def sample_actions(self) -> torch.Tensor:
        self.handle.sample_actions(self.actions)
        return self.actions

    def ready(self):
        """GPU-side ordering of the outputs: the caller's current torch stream waits for everything launched so far on the
        engine's stream (no host synchronisation).  Use before touching output tensors outside `with torch.cuda.stream(r.stream)`."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def synchronize(self):
        """Wait for the engine's stream; raises if a step saw an out-of-range action."""
        try:
            self.handle.sync()
        except _native.MxvError as e:
            if e.code == _native.ERR_INVALID_ACTION:
                raise AssertionError(e.message) from None
            raise

    def close(self):
        self.handle.close()
