"""CPU stand-in for gym_amd.rollout.DeviceRollout, backed by the oracle, for the world_size-2 gloo tests.

Lives under tests/ on purpose: the oracle is test infrastructure and may not be imported by gym_amd/.  It has the
engine protocol ShardedRollout documents (reset / rollout / rollout_per_step / final_tensors / synchronize / stream)
with CPU torch tensors, so the partition rule, the Philox global-index contract and the all-gather plumbing of
gym_amd.distributed / gym_amd.mixed run unchanged over `gloo`."""
import numpy as np
import torch

from gym_amd.registration import spec as _spec
from oracle.oracle import OracleVecEnv


class OracleRollout:
    stream = None

    def __init__(self, id, num_envs, *, env_offset=0, seed=0, action_seed=0, **_):
        self.spec = _spec(id)
        limit = self.spec.max_episode_steps
        self.o = OracleVecEnv(self.spec.kind, num_envs, -1 if limit is None else limit, seed=seed,
                              action_seed=action_seed, env_offset=env_offset)
        self.num_envs = num_envs
        self._last = None

    def reset(self, seed=None):
        obs = torch.from_numpy(self.o.reset(seed=seed))
        n = self.num_envs
        self._last = (obs, torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.uint8),
                      torch.zeros(n, dtype=torch.uint8))
        return obs

    def _step(self):
        a = self.o.sample_actions()
        obs, rew, term, trunc, _, _ = self.o.step(a)
        self._last = (torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(term.astype(np.uint8)),
                      torch.from_numpy(trunc.astype(np.uint8)))
        return a

    def rollout(self, K, **_):
        for _i in range(K):
            self._step()
        return self._last

    def rollout_per_step(self, K, out=None, **_):
        keys = ("obs", "reward", "terminated", "truncated")
        acc = {k: [] for k in keys + ("actions",)}
        for _i in range(K):
            a = self._step()
            for k, t in zip(keys, self._last):
                acc[k].append(t)
            acc["actions"].append(torch.from_numpy(np.asarray(a)))
        return {k: torch.stack(v) for k, v in acc.items()}

    def final_tensors(self):
        return self._last

    def synchronize(self):
        pass

    def close(self):
        pass


class _SnapshotHandle:
    """The one hook of the HIP handle ShardedRollout looks for: a second destination for the final tensors of every rollout."""

    def __init__(self):
        self.views = None

    def set_final_snapshot(self, obs, reward, terminated, truncated):
        self.views = (obs, reward, terminated, truncated)


class OracleRolloutInKernelSnapshot(OracleRollout):
    """OracleRollout with the HIP engine's in-kernel final snapshot (mxv_set_final_snapshot): every rollout also deposits its
    last step's outputs into the armed snapshot set, so ShardedRollout runs the two-set / flip-in-_arm control flow it
    runs on the device — the path where a rollout issued between gather_async() and wait_gather() changes `_cur`."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.handle = _SnapshotHandle()

    def _deposit(self):
        if self.handle.views is not None:
            for dst, src in zip(self.handle.views, self._last):
                dst.copy_(src)

    def rollout(self, K, **kw):
        out = super().rollout(K, **kw)
        self._deposit()
        return out

    def rollout_per_step(self, K, out=None, **kw):
        res = super().rollout_per_step(K, out=out, **kw)
        self._deposit()
        return res


class OracleNormBackend:
    """CPU stand-in for gym_amd.normalize.HipNormBackend (same methods, CPU torch tensors), backed by the oracle's
    restatement of the device definition (oracle/normalize.c, split sums/apply form): lets the product's
    RunningNormalizer run its sharded path (all-gather of the per-step sums) over `gloo`."""
    stream = None
    torch_device = torch.device("cpu")

    def __init__(self, num_envs, obs_dim, gamma=0.99, obs_epsilon=1e-8, rew_epsilon=1e-8):
        from oracle.oracle import RunningNorm

        self.rn = RunningNorm(num_envs, obs_dim, gamma=gamma, obs_epsilon=obs_epsilon, rew_epsilon=rew_epsilon, mode=1)

    def obs_sums(self, K, x, sums):
        sums.copy_(torch.from_numpy(self.rn.obs_sums(x.numpy())))

    def obs_apply(self, K, x, y, epsilon, all_sums, world, total_rows):
        self.rn.obs_epsilon = epsilon
        y.copy_(torch.from_numpy(self.rn.obs_apply(x.numpy(), all_sums.numpy(), total_rows)).to(y.dtype))

    def reward_sums(self, K, reward, terminated, truncated, gamma, sums):
        self.rn.gamma = gamma
        sums.copy_(torch.from_numpy(self.rn.reward_sums(reward.numpy(), terminated.numpy(), truncated.numpy())))

    def reward_apply(self, K, reward, out, epsilon, all_sums, world, total_rows):
        self.rn.rew_epsilon = epsilon
        out.copy_(torch.from_numpy(self.rn.reward_apply(reward.numpy(), all_sums.numpy(), total_rows)).to(out.dtype))

    def obs_state(self):
        return self.rn.obs_mean.copy(), self.rn.obs_var.copy(), float(self.rn.obs_count[0])

    def reward_state(self):
        return self.rn.ret_mean.copy(), self.rn.ret_var.copy(), float(self.rn.ret_count[0]), self.rn.returns.copy()

    def close(self):
        pass


class FakeHandle:
    """Stand-in for gym_amd._native.Handle (the ctypes handle of the HIP engine) backed by the oracle, with just the calls
    HipVectorEnv makes on the copying NumPy path.  Lets the HOST-side adapter (class hierarchy, spaces, errors, infos, wrapper
    compatibility with the reference's own classes) be exercised in the GPU-less build container; never used by the product."""

    def __init__(self, kind, num_envs, max_episode_steps, device=0, env_offset=0, seed=0, action_seed=0, flags=0):
        self.o = OracleVecEnv(kind, num_envs, max_episode_steps, seed=seed, action_seed=action_seed, env_offset=env_offset,
                              autoreset=not (flags & 4))          # MXV_FLAG_NO_AUTORESET
        self.env_id, self.num_envs, self.device, self.flags = kind, num_envs, device, flags
        self.max_episode_steps = int(max_episode_steps)
        self.O, self.S = self.o.O, self.o.S
        self.action_dtype = np.int64 if self.o.discrete else np.float32
        self.closed = False

    def get_params(self):
        return self.o.P.copy()

    def set_params(self, p):
        self.o.P[:] = p

    def seed(self, base_seed, per_env=None):
        self.o.base_seed = int(base_seed) & (2**64 - 1)
        self.o.seeds = None if per_env is None else np.asarray(per_env, dtype=np.uint64).copy()
        self.o.t = self.o.r = 0
        self.o.episodes[:] = 0

    def reset_host(self, mask=None, bounds=None):
        return self.o.reset(mask=mask, bounds=bounds)

    def step_host(self, actions, want_final=True, pooled=False):
        obs, rew, term, trunc, fin, _ = self.o.step(actions)
        return obs, rew, term, trunc, fin

    def step_host_block(self, actions, want_final=True):
        return self.step_host(actions, want_final=want_final)

    def get_state(self):
        return self.o.state.copy(), self.o.elapsed.copy()

    def set_state(self, state=None, elapsed=None):
        if state is not None:
            self.o.state[:] = state
        if elapsed is not None:
            self.o.elapsed[:] = elapsed

    def close(self):
        self.closed = True


class PackedFakeHandle(FakeHandle):
    """FakeHandle that behaves like the handle of a LARGE vector env: final_packed() is supported, host steps leave the final
    observations (and, with episode statistics on, returns and lengths) of the finished envs as packed (index, row) records in
    ascending env order and return no dense final array; the resets of the autoreset come from the oracle.  Exercises the
    adapter's packed code paths (LazyInfos built from packed rows, RecordEpisodeStatistics._step_packed) without a device."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        from oracle.oracle import EpisodeStats

        self._packed_on = False
        self._stats = None
        self._EpisodeStats = EpisodeStats
        self._rec = (np.zeros(0, np.int32), np.zeros((0, self.O), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32))
        self._dense_stats = None

    def final_packed(self, enable=True):
        self._packed_on = bool(enable)
        return self._packed_on

    def episode_stats(self, enable=True):
        self._stats = self._EpisodeStats(self.num_envs) if enable else None

    def reset_host(self, mask=None, bounds=None):
        if self._stats is not None and mask is None:
            self._stats.reset()
        return super().reset_host(mask=mask, bounds=bounds)

    def step_host_block(self, actions, want_final=True):
        obs, rew, term, trunc, fin, _ = self.o.step(actions)
        done = term | trunc
        idx = np.flatnonzero(done).astype(np.int32)
        r = l = None
        if self._stats is not None:
            r, l, _ = self._stats.step(rew, term, trunc)
            self._dense_stats = (r, l)
        self._rec = (idx, fin[idx].copy(), None if r is None else r[idx].copy(), None if l is None else l[idx].copy())
        return obs, rew, term, trunc, (None if self._packed_on else fin)

    def final_packed_rows(self):
        return self._rec[0].copy(), self._rec[1].copy()

    def final_packed_stats(self):
        return self._rec[0].copy(), self._rec[2].copy(), self._rec[3].copy()

    def episode_stats_host(self, want_running=False):
        r, l = self._dense_stats
        return (r, l, self._stats.returns.copy()) if want_running else (r, l)


class _FakeStats:
    """The fused episode statistics of the toy_text handles, restated with the oracle's EpisodeStats (float32 return accumulator; the
    length is the wrapper's own counter, which equals the TimeLimit counter the kernels report)."""

    def _stats_init(self):
        self._stats = None
        self._last = None

    def episode_stats(self, enable=True):
        from oracle.oracle import EpisodeStats

        self._stats = EpisodeStats(self.num_envs) if enable else None
        self._stats_on = bool(enable)

    def _stats_step(self, rew, term, trunc):
        if self._stats is not None:
            r, l, _ = self._stats.step(rew, term, trunc)
            self._last = (r, l)

    def episode_stats_host(self, want_running=False):
        r, l = self._last if self._last is not None else (np.zeros(self.num_envs, np.float32), np.zeros(self.num_envs, np.int32))
        return (r.copy(), l.copy(), self._stats.returns.copy()) if want_running else (r.copy(), l.copy())

    def set_running_returns(self, running):
        self._stats.returns[:] = np.asarray(running, np.float32)


class FakeTab(_FakeStats):
    """Stand-in for gym_amd._native.Tab backed by the oracle's tabular twin (OracleTabEnv): the calls HipTabularVectorEnv and the
    statistics / normalisation wrappers make on the NumPy path.  Same purpose and same rule as FakeHandle: host logic on the GPU-less
    box, never used by the product."""

    def __init__(self, num_states, num_actions, cum_prob, prob, next_state, reward, terminated, initial_cum, num_envs, max_episode_steps,
                 *, device=0, env_offset=0, seed=0, action_seed=0, compact=False, general_kernel=False):
        from oracle.oracle import OracleTabEnv

        self.o = OracleTabEnv(cum_prob, prob, next_state, reward, terminated, initial_cum, num_envs, max_episode_steps, seed=seed,
                              action_seed=action_seed, env_offset=env_offset)
        self.S, self.A, self.M, self.num_envs, self.device, self.compact = self.o.S, self.o.A, self.o.M, int(num_envs), int(device), bool(compact)
        self._stats_init()

    def seed(self, base_seed, per_env_seeds=None):
        self.o.base_seed = int(base_seed) & (2**64 - 1)
        self.o.seeds = None if per_env_seeds is None else np.asarray(per_env_seeds, dtype=np.uint64).copy()
        self.o.t = self.o.r = 0

    def seed_actions(self, action_seed):
        self.o.action_seed = int(action_seed) & (2**64 - 1)

    def reset_host(self, mask=None):
        if self._stats is not None and mask is None:
            self._stats.reset()
        return self.o.reset(mask=mask)

    def step_host(self, actions, uniforms=None, pooled=False):
        from gym_amd import _native

        try:
            out = self.o.step(actions, uniforms)
        except KeyError as e:
            raise _native.MxvError(_native.ERR_INVALID_ACTION, str(e)) from None
        self._stats_step(out["reward"], out["terminated"], out["truncated"])
        return out["obs"], out["reward"], out["terminated"], out["truncated"], out["prob"], out["final_obs"], out["final_prob"]

    def get_state(self):
        return self.o.state.copy(), self.o.elapsed.copy()

    def set_state(self, state=None, elapsed=None):
        if state is not None:
            self.o.state[:] = state
        if elapsed is not None:
            self.o.elapsed[:] = elapsed

    def get_counters(self):
        return self.o.t, self.o.r

    def set_counters(self, t, r):
        self.o.t, self.o.r = int(t), int(r)

    def snapshot(self):
        on = self._stats is not None
        return dict(num_envs=self.num_envs, state=self.o.state.copy(), elapsed=self.o.elapsed.copy(), t=self.o.t, r=self.o.r, base_seed=self.o.base_seed,
                    per_env_seeds=None if self.o.seeds is None else self.o.seeds.copy(), action_seed=self.o.action_seed, stats_on=on,
                    running_returns=self._stats.returns.copy() if on else None)

    def restore(self, snap):
        if snap["num_envs"] != self.num_envs:
            raise ValueError("snapshot does not fit this handle")
        self.seed(snap["base_seed"], snap["per_env_seeds"])
        self.seed_actions(snap["action_seed"])
        self.set_state(snap["state"], snap["elapsed"])
        self.set_counters(snap["t"], snap["r"])
        if snap.get("stats_on"):
            self.episode_stats(True)
            self.set_running_returns(snap["running_returns"])

    def close(self):
        pass


class FakeBlackjack(_FakeStats):
    """Stand-in for gym_amd._native.Blackjack backed by OracleBlackjack (the hands as card lists like the reference keeps them; the
    snapshot carries those arrays instead of the device's packed words).  Host logic only, like FakeTab."""

    def __init__(self, num_envs, *, natural=False, sab=False, max_episode_steps=-1, device=0, env_offset=0, seed=0, action_seed=0):
        from oracle.oracle import OracleBlackjack

        self.o = OracleBlackjack(num_envs, natural=natural, sab=sab, max_episode_steps=max_episode_steps, seed=seed, action_seed=action_seed,
                                 env_offset=env_offset)
        self.num_envs, self.device = int(num_envs), int(device)
        self._stats_init()

    def seed(self, base_seed, per_env_seeds=None, action_seed=0):
        self.o.base_seed = int(base_seed) & (2**64 - 1)
        self.o.seeds = None if per_env_seeds is None else np.asarray(per_env_seeds, dtype=np.uint64).copy()
        self.o.action_seed = int(action_seed) & (2**64 - 1)
        self.o.t = self.o.r = 0

    def reset_host(self, cards=None):
        if self._stats is not None:
            self._stats.reset()
        return self.o.reset(cards=cards)

    def step_host(self, actions, cards=None, pooled=False):
        from gym_amd import _native

        try:
            out = self.o.step(actions, cards)
        except AssertionError as e:
            raise _native.MxvError(_native.ERR_INVALID_ACTION, str(e)) from None
        self._stats_step(out["reward"], out["terminated"], out["truncated"])
        return out["obs"], out["reward"], out["terminated"], out["truncated"], out["final_obs"]

    def snapshot(self):
        from gym_amd import _native

        on = self._stats is not None
        return dict(num_envs=self.num_envs, state=(self.o.dealer.copy(), self.o.player.copy()), elapsed=self.o.elapsed.copy(), t=self.o.t, r=self.o.r,
                    base_seed=self.o.base_seed, per_env_seeds=None if self.o.seeds is None else self.o.seeds.copy(), action_seed=self.o.action_seed,
                    draw_contract=_native.BJ_DRAW_CONTRACT, stats_on=on, running_returns=self._stats.returns.copy() if on else None)

    def restore(self, snap):
        from gym_amd import _native

        if snap.get("draw_contract", 1) != _native.BJ_DRAW_CONTRACT:
            raise ValueError("Blackjack snapshot was taken under another draw contract")
        self.seed(snap["base_seed"], snap["per_env_seeds"], snap["action_seed"])
        self.o.dealer[:], self.o.player[:] = snap["state"]
        self.o.elapsed[:] = snap["elapsed"]
        self.o.t, self.o.r = int(snap["t"]), int(snap["r"])
        if snap.get("stats_on"):
            self.episode_stats(True)
            self.set_running_returns(snap["running_returns"])

    def close(self):
        pass
