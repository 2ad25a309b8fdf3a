"""ctypes front-end of the CPU oracle (oracle/classic_control.c).

TEST INFRASTRUCTURE ONLY — see the header of classic_control.c.  Importable only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; gym_amd/ never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liborc.so")

ENV_IDS = {"CartPole": 0, "Pendulum": 1, "Acrobot": 2, "MountainCar": 3, "MountainCarContinuous": 4}
MAX_PARAMS = 12
DISCRETE = {0: 2, 2: 3, 3: 3}  # env_id -> number of actions


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Returns the library path."""
    srcs = [os.path.join(_HERE, f) for f in ("classic_control.c", "normalize.c", "tabular.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u64, i64, i32, u32 = C.c_uint64, C.c_int64, C.c_int32, C.c_uint32
        vp = C.c_void_p
        L.orc_state_dim.restype = C.c_int
        L.orc_obs_dim.restype = C.c_int
        L.orc_default_params.argtypes = [C.c_int, vp]
        L.orc_default_reset_bounds.argtypes = [C.c_int, vp]
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_sample_actions.argtypes = [C.c_int, i64, u64, u64, u64, vp, vp, vp]
        L.orc_vec_reset.argtypes = [C.c_int, i64, u64, vp, u64, vp, vp, vp, vp, vp, vp]
        L.orc_vec_step.argtypes = [C.c_int, i64, u64, vp, C.c_int, C.c_int, vp, u64, u64, vp, vp, vp, vp,
                                   vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_vec_step.restype = i64
        L.orc_vec_step_beyond.argtypes = [C.c_int, i64, u64, vp, C.c_int, C.c_int, vp, u64, u64, vp, vp, vp, vp,
                                   vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_vec_step_beyond.restype = i64
        L.orc_rollout.argtypes = [C.c_int, i64, u64, vp, C.c_int, u64, u64, u64, vp, C.c_int, vp, vp, vp,
                                  vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_norm_obs_batches.argtypes = [vp, vp, vp, C.c_double, vp, i64, i64, C.c_int, C.c_int, vp]
        L.orc_norm_reward_steps.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp, i64, i64, C.c_int, vp]
        L.orc_subnorm_obs.argtypes = [vp, vp, vp, C.c_double, vp, vp, vp, vp, i64, i64, C.c_int, vp, vp]
        L.orc_subnorm_rew.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp, i64, i64, vp]
        L.orc_tab_reset.argtypes = [C.c_int, vp, i64, u64, vp, u64, u64, u32, vp, vp, vp, vp]
        L.orc_tab_step.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, i64, u64, vp, u64, u64, u64, C.c_int,
                                   vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_tab_step.restype = i64
        L.orc_bj_reset.argtypes = [i64, u64, vp, u64, u64, u32, vp, vp, vp, vp, vp]
        L.orc_bj_step.argtypes = [i64, u64, vp, u64, u64, u64, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp,
                                  vp, vp, vp, vp, vp]
        L.orc_bj_step.restype = i64
        L.orc_bj_stream_cards.argtypes = [i64, u64, u64, C.c_int, vp]
        L.orc_norm_obs_sums.argtypes = [vp, i64, i64, C.c_int, vp]
        L.orc_norm_obs_apply.argtypes = [vp, vp, vp, C.c_double, vp, i64, i64, C.c_int, vp, C.c_int, i64, vp]
        L.orc_norm_reward_sums.argtypes = [vp, vp, vp, vp, i64, i64, C.c_double, vp]
        L.orc_norm_reward_apply.argtypes = [vp, vp, vp, C.c_double, vp, i64, i64, vp, C.c_int, i64, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def default_params(env_id: int) -> np.ndarray:
    P = np.zeros(MAX_PARAMS, dtype=np.float64)
    lib().orc_default_params(env_id, _p(P))
    return P


def default_reset_bounds(env_id: int) -> np.ndarray:
    b = np.zeros(2, dtype=np.float64)
    lib().orc_default_reset_bounds(env_id, _p(b))
    return b


class OracleVecEnv:
    """Batched CPU twin of the device engine: same state layout (SoA fp64), same RNG contract,
    same step/reset/autoreset semantics as SyncVectorEnv over TimeLimit-wrapped classic-control envs."""

    def __init__(self, env_id: int, num_envs: int, max_episode_steps: int, seed: int = 0,
                 action_seed: int = 0, env_offset: int = 0, params=None, autoreset: bool = True):
        self.env_id = int(env_id)
        self.n = int(num_envs)
        self.S = lib().orc_state_dim(self.env_id)
        self.O = lib().orc_obs_dim(self.env_id)
        self.max_episode_steps = int(max_episode_steps)
        self.base_seed = int(seed) & (2**64 - 1)
        self.action_seed = int(action_seed) & (2**64 - 1)
        self.env0 = int(env_offset)
        self.autoreset = bool(autoreset)
        self.P = default_params(self.env_id) if params is None else np.array(params, dtype=np.float64)
        self.bounds = default_reset_bounds(self.env_id)
        self.seeds = None  # optional per-env uint64 seeds
        self.state = np.zeros((self.S, self.n), dtype=np.float64)
        self.elapsed = np.zeros(self.n, dtype=np.int32)
        self.t = 0  # vector-step index since seeding
        self.r = 0  # explicit reset calls since seeding (informational)
        self.episodes = np.zeros(self.n, dtype=np.uint32)  # per-env reset ordinals = position of each env's reset stream
        self.discrete = self.env_id in DISCRETE
        # CartPole without autoreset: which envs have terminated before (cartpole.py:169-184, steps_beyond_terminated is not None)
        self.beyond = np.zeros(self.n, dtype=np.uint8) if (self.env_id == 0 and not self.autoreset) else None

    # -- RNG-contract helpers ---------------------------------------------------------
    def sample_actions(self, t=None):
        t = self.t if t is None else t
        ai = np.zeros(self.n, dtype=np.int64)
        af = np.zeros(self.n, dtype=np.float32)
        lib().orc_sample_actions(self.env_id, self.n, self.env0, self.action_seed, t, _p(self.P), _p(ai), _p(af))
        return ai if self.discrete else af

    def reset(self, seed=None, mask=None, bounds=None):
        """Explicit reset. seed: None (continue streams), int (base seed) or array of per-env seeds."""
        if seed is not None:
            if np.ndim(seed) == 0:
                self.base_seed = int(seed) & (2**64 - 1)
                self.seeds = None
            else:
                self.seeds = np.asarray(seed, dtype=np.uint64).copy()
            self.t = 0
            self.r = 0
            self.episodes[:] = 0
        self.r += 1
        b = self.bounds if bounds is None else np.asarray(bounds, dtype=np.float64)
        obs = np.zeros((self.n, self.O), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_vec_reset(self.env_id, self.n, self.env0, _p(self.seeds), self.base_seed, _p(self.episodes),
                            _p(b), _p(m), _p(self.state), _p(self.elapsed), _p(obs))
        if self.beyond is not None:                     # steps_beyond_terminated = None (cartpole.py:205)
            self.beyond[slice(None) if m is None else m.astype(bool)] = 0
        return obs

    def step(self, actions):
        n, O = self.n, self.O
        obs = np.zeros((n, O), dtype=np.float32)
        reward = np.zeros(n, dtype=np.float64)
        term = np.zeros(n, dtype=np.uint8)
        trunc = np.zeros(n, dtype=np.uint8)
        final_obs = np.zeros((n, O), dtype=np.float32)
        final_mask = np.zeros(n, dtype=np.uint8)
        if self.discrete:
            ai = np.ascontiguousarray(actions, dtype=np.int64).reshape(n)
            af = None
        else:
            af = np.ascontiguousarray(actions, dtype=np.float32).reshape(n)
            ai = None
        bad = lib().orc_vec_step_beyond(self.env_id, n, self.env0, _p(self.P), self.max_episode_steps,
                                        int(self.autoreset), _p(self.seeds), self.base_seed, self.t, _p(self.episodes), _p(self.bounds),
                                        _p(ai), _p(af), _p(self.state), _p(self.elapsed), _p(obs), _p(reward),
                                        _p(term), _p(trunc), _p(final_obs), _p(final_mask), _p(self.beyond))
        if bad:
            raise AssertionError(f"{bad} invalid discrete action(s)")
        self.t += 1
        return obs, reward, term.astype(bool), trunc.astype(bool), final_obs, final_mask.astype(bool)

    def rollout(self, K: int):
        """K random-action steps (Philox action stream). Returns (sum_reward, num_done)."""
        n, O = self.n, self.O
        ai = np.zeros(n, dtype=np.int64)
        af = np.zeros(n, dtype=np.float32)
        obs = np.zeros((n, O), dtype=np.float32)
        reward = np.zeros(n, dtype=np.float64)
        term = np.zeros(n, dtype=np.uint8)
        trunc = np.zeros(n, dtype=np.uint8)
        sr = C.c_double(0.0)
        nd = C.c_int64(0)
        lib().orc_rollout(self.env_id, n, self.env0, _p(self.P), self.max_episode_steps, self.base_seed,
                          self.action_seed, self.t, _p(self.episodes), int(K), _p(self.bounds), _p(self.state), _p(self.elapsed),
                          _p(ai), _p(af), _p(obs), _p(reward), _p(term), _p(trunc), C.byref(sr), C.byref(nd))
        self.t += int(K)
        return sr.value, nd.value, obs, reward, term.astype(bool), trunc.astype(bool)


class EpisodeStats:
    """gym.wrappers.RecordEpisodeStatistics restated for a batched stream (gym/wrappers/record_episode_statistics.py:
    89-151): float32 cumulative returns (`episode_returns += rewards` casts the float64 sum back to float32), int32
    lengths, both zeroed where an episode ended.  Checked against the reference's own wrapper in tests/golden."""

    def __init__(self, n: int):
        self.returns = np.zeros(n, dtype=np.float32)   # :92
        self.lengths = np.zeros(n, dtype=np.int32)     # :93

    def reset(self):
        self.returns[:] = 0
        self.lengths[:] = 0

    def step(self, rewards, terminated, truncated):
        """-> (ep_return f32[N], ep_length i32[N], mask bool[N]); entries outside the mask are 0."""
        self.returns += rewards                         # :119
        self.lengths += 1                               # :120
        done = np.asarray(terminated, dtype=bool) | np.asarray(truncated, dtype=bool)
        r = np.where(done, self.returns, 0).astype(np.float32)
        l = np.where(done, self.lengths, 0).astype(np.int32)
        self.returns[done] = 0                          # :142
        self.lengths[done] = 0                          # :143
        return r, l, done


class SubEnvNorm:
    """N independent RunningMeanStd objects fed with batches of one row: what `gym.vector.make(id, n, wrappers=[NormalizeObservation,
    NormalizeReward])` keeps (oracle/normalize.c: orc_subnorm_*, following gym/wrappers/normalize.py:17-47,72-93,127-145 under
    gym/vector/sync_vector_env.py:142-156).  The checker of the mxv_subnorm_* kernels; same call shapes as gym_amd._native.SubNorm
    with NumPy arrays in place of device tensors."""

    def __init__(self, dim: int, num_envs: int, **_):
        self.dim, self.num_envs = int(dim), int(num_envs)
        self.mean = np.zeros((self.num_envs, self.dim), np.float64)
        self.var = np.ones((self.num_envs, self.dim), np.float64)
        self.count = np.full(self.num_envs, 1e-4, np.float64)
        self.returns = np.zeros(self.num_envs, np.float64)

    def observations(self, K, x, fin, te, tr, y, out_f32, yfin, epsilon):
        """x / fin float32 [K][n][dim], te / tr uint8 [K][n] or None; y float32 / float64 [K][n][dim] (written), yfin float64 or None."""
        n, D = self.num_envs, self.dim
        xs = np.ascontiguousarray(x, dtype=np.float32).reshape(K, n, D)
        fs = None if fin is None else np.ascontiguousarray(fin, dtype=np.float32).reshape(K, n, D)
        tes = None if te is None else np.ascontiguousarray(np.asarray(te).reshape(K, n), dtype=np.uint8)
        trs = None if tr is None else np.ascontiguousarray(np.asarray(tr).reshape(K, n), dtype=np.uint8)
        y64 = np.zeros((K, n, D), np.float64)
        yf = None if yfin is None else np.zeros((K, n, D), np.float64)
        lib().orc_subnorm_obs(_p(self.mean), _p(self.var), _p(self.count), float(epsilon), _p(xs), _p(fs), _p(tes), _p(trs), K, n, D,
                              _p(y64), _p(yf))
        np.copyto(np.asarray(y).reshape(K, n, D), y64, casting="same_kind")     # float64 -> float32 rounds once, like np.stack into the buffer
        if yfin is not None:
            np.asarray(yfin).reshape(K, n, D)[...] = yf
        return y

    def rewards(self, K, rew, reward_f32, te, tr, out, gamma, epsilon):
        n = self.num_envs
        rs = np.ascontiguousarray(rew, dtype=np.float64).reshape(K, n)
        tes = np.ascontiguousarray(np.asarray(te).reshape(K, n), dtype=np.uint8)
        trs = np.ascontiguousarray(np.asarray(tr).reshape(K, n), dtype=np.uint8)
        o = np.zeros((K, n), np.float64)
        lib().orc_subnorm_rew(_p(self.returns), _p(self.mean), _p(self.var), _p(self.count), float(gamma), float(epsilon), _p(rs), _p(tes),
                              _p(trs), K, n, _p(o))
        np.copyto(np.asarray(out).reshape(K, n), o, casting="same_kind")
        return out

    def get_state(self):
        return self.mean.copy(), self.var.copy(), self.count.copy(), self.returns.copy()

    def set_state(self, mean, var, count, returns=None):
        self.mean[...] = np.asarray(mean).reshape(self.mean.shape)
        self.var[...] = np.asarray(var).reshape(self.var.shape)
        self.count[...] = count
        if returns is not None:
            self.returns[...] = returns

    def close(self):
        pass


class RunningNorm:
    """gym.wrappers.NormalizeObservation / NormalizeReward restated for a batched stream (oracle/normalize.c follows
    gym/wrappers/normalize.py:8-144).  mode 0 = the reference's own arithmetic (float32 row-by-row batch moments, NumPy
    pairwise sums) — pinned bit-exact by tests/golden/normalize_*.npz; mode 1 = exact batch sums rounded to the
    reference's moment dtype, the definition the device kernels implement."""

    def __init__(self, num_envs: int, obs_dim: int, gamma: float = 0.99, obs_epsilon: float = 1e-8,
                 rew_epsilon: float = 1e-8, mode: int = 0):
        self.n, self.O, self.mode = int(num_envs), int(obs_dim), int(mode)
        self.gamma, self.obs_epsilon, self.rew_epsilon = float(gamma), float(obs_epsilon), float(rew_epsilon)
        self.obs_mean = np.zeros(self.O, np.float64)      # RunningMeanStd.__init__, normalize.py:12-15
        self.obs_var = np.ones(self.O, np.float64)
        self.obs_count = np.array([1e-4], np.float64)
        self.ret_mean = np.zeros(1, np.float64)
        self.ret_var = np.ones(1, np.float64)
        self.ret_count = np.array([1e-4], np.float64)
        self.returns = np.zeros(self.n, np.float64)       # :123

    def normalize_obs(self, x):
        """x float32 [K][n][O] (or [n][O]) -> float64, K consecutive NormalizeObservation.normalize calls."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        xs = x.reshape(-1, self.n, self.O)
        y = np.zeros(xs.shape, np.float64)
        lib().orc_norm_obs_batches(_p(self.obs_mean), _p(self.obs_var), _p(self.obs_count), self.obs_epsilon, _p(xs),
                                   xs.shape[0], self.n, self.O, self.mode, _p(y))
        return y.reshape(x.shape)

    def normalize_rewards(self, rew, terminated, truncated):
        """rew float64 [K][n] (or [n]) + flags -> float64, K consecutive NormalizeReward.step calls."""
        r = np.ascontiguousarray(rew, dtype=np.float64)
        rs = r.reshape(-1, self.n)
        te = np.ascontiguousarray(np.asarray(terminated).reshape(rs.shape), dtype=np.uint8)
        tr = np.ascontiguousarray(np.asarray(truncated).reshape(rs.shape), dtype=np.uint8)
        out = np.zeros(rs.shape, np.float64)
        lib().orc_norm_reward_steps(_p(self.returns), _p(self.ret_mean), _p(self.ret_var), _p(self.ret_count), self.gamma,
                                    self.rew_epsilon, _p(rs), _p(te), _p(tr), rs.shape[0], self.n, self.mode, _p(out))
        return out.reshape(r.shape)

    # -- split form (a shard of a vector env sharded over `world` ranks), mode 1 arithmetic -------------------------------
    def obs_sums(self, x):
        xs = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.n, self.O)
        sums = np.zeros((xs.shape[0], 2 * self.O), np.float64)
        lib().orc_norm_obs_sums(_p(xs), xs.shape[0], self.n, self.O, _p(sums))
        return sums

    def obs_apply(self, x, all_sums, total_rows):
        x = np.ascontiguousarray(x, dtype=np.float32)
        xs = x.reshape(-1, self.n, self.O)
        a = np.ascontiguousarray(all_sums, dtype=np.float64)
        y = np.zeros(xs.shape, np.float64)
        lib().orc_norm_obs_apply(_p(self.obs_mean), _p(self.obs_var), _p(self.obs_count), self.obs_epsilon, _p(xs),
                                 xs.shape[0], self.n, self.O, _p(a), a.shape[0], int(total_rows), _p(y))
        return y.reshape(x.shape)

    def reward_sums(self, rew, terminated, truncated):
        rs = np.ascontiguousarray(rew, dtype=np.float64).reshape(-1, self.n)
        te = np.ascontiguousarray(np.asarray(terminated).reshape(rs.shape), dtype=np.uint8)
        tr = np.ascontiguousarray(np.asarray(truncated).reshape(rs.shape), dtype=np.uint8)
        sums = np.zeros((rs.shape[0], 2), np.float64)
        lib().orc_norm_reward_sums(_p(self.returns), _p(rs), _p(te), _p(tr), rs.shape[0], self.n, self.gamma, _p(sums))
        return sums

    def reward_apply(self, rew, all_sums, total_rows):
        r = np.ascontiguousarray(rew, dtype=np.float64)
        rs = r.reshape(-1, self.n)
        a = np.ascontiguousarray(all_sums, dtype=np.float64)
        out = np.zeros(rs.shape, np.float64)
        lib().orc_norm_reward_apply(_p(self.ret_mean), _p(self.ret_var), _p(self.ret_count), self.rew_epsilon, _p(rs),
                                    rs.shape[0], self.n, _p(a), a.shape[0], int(total_rows), _p(out))
        return out.reshape(r.shape)


class OracleTabEnv:
    """Batched CPU twin of the mxv_tab engine (oracle/tabular.c): tabular toy_text env under SyncVectorEnv + TimeLimit.
    Tables as in include/mxv.h: cum_prob / prob / next_state / reward / terminated [S][A][M], initial_cum [S]."""

    def __init__(self, cum_prob, prob, next_state, reward, terminated, initial_cum, num_envs, max_episode_steps,
                 seed=0, action_seed=0, env_offset=0):
        self.cum = np.ascontiguousarray(cum_prob, np.float64)
        self.S, self.A, self.M = self.cum.shape
        self.prob = np.ascontiguousarray(prob, np.float64)
        self.next = np.ascontiguousarray(next_state, np.int32)
        self.reward = np.ascontiguousarray(reward, np.float64)
        self.term_tab = np.ascontiguousarray(terminated, np.uint8)
        self.init_cum = np.ascontiguousarray(initial_cum, np.float64)
        self.n = int(num_envs)
        self.max_episode_steps = -1 if max_episode_steps is None else int(max_episode_steps)
        self.base_seed = int(seed) & (2**64 - 1)
        self.action_seed = int(action_seed) & (2**64 - 1)
        self.env0 = int(env_offset)
        self.seeds = None
        self.state = np.zeros(self.n, np.int32)
        self.elapsed = np.zeros(self.n, np.int32)
        self.t = 0
        self.r = 0

    def reset(self, seed=None, mask=None):
        if seed is not None:
            if np.ndim(seed) == 0:
                self.base_seed, self.seeds = int(seed) & (2**64 - 1), None
            else:
                self.seeds = np.asarray(seed, dtype=np.uint64).copy()
            self.t = 0
            self.r = 0
        self.r += 1
        obs = np.zeros(self.n, np.int64)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_tab_reset(self.S, _p(self.init_cum), self.n, self.env0, _p(self.seeds), self.base_seed, self.t, self.r,
                            _p(m), _p(self.state), _p(self.elapsed), _p(obs))
        return obs

    def step(self, actions=None, uniforms=None):
        """-> dict(actions, obs, reward, terminated, truncated, prob, final_obs, final_prob, final_mask)"""
        n = self.n
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.int64).reshape(n)
        u = None if uniforms is None else np.ascontiguousarray(uniforms, dtype=np.float64).reshape(2, n)
        out = dict(actions=np.zeros(n, np.int64), obs=np.zeros(n, np.int64), reward=np.zeros(n, np.float64),
                   terminated=np.zeros(n, np.uint8), truncated=np.zeros(n, np.uint8), prob=np.zeros(n, np.float64),
                   final_obs=np.zeros(n, np.int64), final_prob=np.zeros(n, np.float64), final_mask=np.zeros(n, np.uint8))
        bad = lib().orc_tab_step(self.S, self.A, self.M, _p(self.cum), _p(self.prob), _p(self.next), _p(self.reward),
                                 _p(self.term_tab), _p(self.init_cum), n, self.env0, _p(self.seeds), self.base_seed,
                                 self.action_seed, self.t, self.max_episode_steps, _p(a), _p(u), _p(self.state),
                                 _p(self.elapsed), _p(out["actions"]), _p(out["obs"]), _p(out["reward"]),
                                 _p(out["terminated"]), _p(out["truncated"]), _p(out["prob"]), _p(out["final_obs"]),
                                 _p(out["final_prob"]), _p(out["final_mask"]))
        if bad:
            raise KeyError(f"{bad} invalid action(s)")
        self.t += 1
        for k in ("terminated", "truncated", "final_mask"):
            out[k] = out[k].astype(bool)
        return out


class OracleBlackjack:
    """Batched CPU twin of the mxv_bj engine (oracle/tabular.c, Blackjack-v1 under SyncVectorEnv): hands kept as card lists
    like the reference; cards injected (the reference's own np_random.choice draws) or from the Philox draw stream."""

    MAX_DRAWS = 24

    def __init__(self, num_envs, natural=False, sab=True, max_episode_steps=-1, seed=0, action_seed=0, env_offset=0):
        self.n = int(num_envs)
        self.natural, self.sab = int(bool(natural)), int(bool(sab))
        self.max_episode_steps = -1 if max_episode_steps is None else int(max_episode_steps)
        self.base_seed = int(seed) & (2**64 - 1)
        self.action_seed = int(action_seed) & (2**64 - 1)
        self.env0 = int(env_offset)
        self.seeds = None
        self.dealer = np.zeros((self.n, 33), np.int32)
        self.player = np.zeros((self.n, 33), np.int32)
        self.elapsed = np.zeros(self.n, np.int32)
        self.t = 0
        self.r = 0

    def reset(self, seed=None, cards=None):
        if seed is not None:
            self.base_seed, self.t, self.r = int(seed) & (2**64 - 1), 0, 0
        self.r += 1
        obs = np.zeros((3, self.n), np.int64)
        c = None if cards is None else np.ascontiguousarray(cards, dtype=np.int8).reshape(self.n, 4)
        lib().orc_bj_reset(self.n, self.env0, _p(self.seeds), self.base_seed, self.t, self.r, _p(c), _p(self.dealer),
                           _p(self.player), _p(self.elapsed), _p(obs))
        return obs

    def step(self, actions=None, cards=None):
        n = self.n
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.int64).reshape(n)
        c = None if cards is None else np.ascontiguousarray(cards, dtype=np.int8).reshape(n, self.MAX_DRAWS)
        out = dict(actions=np.zeros(n, np.int64), obs=np.zeros((3, n), np.int64), reward=np.zeros(n), terminated=np.zeros(n, np.uint8),
                   truncated=np.zeros(n, np.uint8), final_obs=np.zeros((3, n), np.int64), final_mask=np.zeros(n, np.uint8))
        bad = lib().orc_bj_step(n, self.env0, _p(self.seeds), self.base_seed, self.action_seed, self.t, self.natural, self.sab,
                                self.max_episode_steps, _p(a), _p(c), self.MAX_DRAWS, _p(self.dealer), _p(self.player),
                                _p(self.elapsed), _p(out["actions"]), _p(out["obs"]), _p(out["reward"]), _p(out["terminated"]),
                                _p(out["truncated"]), _p(out["final_obs"]), _p(out["final_mask"]))
        if bad:
            raise AssertionError(f"{bad} invalid action(s)")
        self.t += 1
        for k in ("terminated", "truncated", "final_mask"):
            out[k] = out[k].astype(bool)
        return out
