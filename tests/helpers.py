"""Shared parity machinery: golden-vector replay and oracle-vs-engine comparison.

An "engine" here is anything with
    set_state(state[S,N] f64, elapsed[N] i32) / get_state() -> (state, elapsed)
    step(actions) -> obs[N,O] f32, reward[N] f64, terminated[N] bool, truncated[N] bool, final_obs[N,O] f32
The CPU oracle (oracle/oracle.py) and the HIP engine behind the C ABI (gym_amd._native.Handle) are both wrapped
to that shape so the very same checks run against either.
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENV_NAMES = ["CartPole", "Pendulum", "Acrobot", "MountainCar", "MountainCarContinuous"]
ENV_IDS = {n: i for i, n in enumerate(ENV_NAMES)}
GYM_IDS = {"CartPole": "CartPole-v1", "Pendulum": "Pendulum-v1", "Acrobot": "Acrobot-v1",
           "MountainCar": "MountainCar-v0", "MountainCarContinuous": "MountainCarContinuous-v0"}
LIMITS = {"CartPole": 500, "Pendulum": 200, "Acrobot": 500, "MountainCar": 200, "MountainCarContinuous": 999}
DISCRETE = {"CartPole": 2, "Pendulum": 0, "Acrobot": 3, "MountainCar": 3, "MountainCarContinuous": 0}


def load_golden(name, kind):
    return np.load(os.path.join(GOLDEN, f"{name}_{kind}.npz"))


def ulps32(a, b):
    """Distance in float32 units-in-the-last-place (0 = bit-identical up to the sign of zero)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


class OracleEngine:
    def __init__(self, name, n, max_steps, autoreset=True, seed=0, action_seed=0, env_offset=0):
        from oracle.oracle import OracleVecEnv

        self.o = OracleVecEnv(ENV_IDS[name], n, max_steps, seed=seed, action_seed=action_seed, env_offset=env_offset,
                              autoreset=autoreset)

    def set_state(self, state, elapsed):
        self.o.state[:] = state
        self.o.elapsed[:] = elapsed
        if self.o.beyond is not None:
            self.o.beyond[:] = 0           # a fresh state: steps_beyond_terminated = None (what make_golden.set_state does to the reference)

    def get_state(self):
        return self.o.state.copy(), self.o.elapsed.copy()

    def step(self, actions):
        obs, rew, term, trunc, fin, fmask = self.o.step(actions)
        return obs, rew, term, trunc, fin

    def set_params(self, p):
        self.o.P[:] = p


class HipEngine:
    """The product path: every call goes through the C ABI (ctypes -> libmxv.so -> HIP kernels)."""

    def __init__(self, name, n, max_steps, autoreset=True, seed=0, action_seed=0, env_offset=0):
        from gym_amd import _native

        flags = 0 if autoreset else _native.FLAG_NO_AUTORESET
        self.h = _native.Handle(ENV_IDS[name], n, max_steps, seed=seed, action_seed=action_seed,
                                env_offset=env_offset, flags=flags)

    def set_state(self, state, elapsed):
        self.h.set_state(state, elapsed)

    def get_state(self):
        return self.h.get_state()

    def step(self, actions):
        return self.h.step_host(actions, want_final=True)

    def set_params(self, p):
        self.h.set_params(p)


# Tolerances of engine-vs-reference comparisons.  Integer/boolean outputs (terminated, truncated, elapsed,
# final mask) are always exact.  strict=True is the oracle's bar (bit-exact against the reference).  The HIP engine
# computes in fp64 like the reference but its sin/cos (ocml) and x*x (vs libm pow) may differ from glibc in the
# last fp64 bit, so observations are held to north_star's rtol=1e-5 AND to <= MAX_OBS_ULPS float32 ulps.
OBS_RTOL = 1e-5
MAX_OBS_ULPS = 2
STATE_RTOL, STATE_ATOL = 1e-12, 1e-13
# Rewards are fp64.  Pendulum's cost holds one float32 term, 0.001*(u**2) (pendulum.py:129): the reference gets u**2
# from libm powf (<= 0.82 ulp, not correctly rounded) while the engine multiplies u*u (correctly rounded), so that
# term (<= 0.004) may differ by one float32 ulp = 2^-31 absolute (measured: libm powf(u,2) != u*u for 0.08 % of
# random u): Pendulum rewards get atol 1e-9 on top of rtol 1e-13; every other env rtol 1e-13 only.
REWARD_RTOL = 1e-13
REWARD_ATOL = {"Pendulum": 1e-9}
REWARD_ATOL_DEFAULT = 1e-300


def compare_step(tag, got, ref, done_ref, strict):
    """got/ref: dicts with obs, reward, terminated, truncated, final_obs, state, elapsed (post-step)."""
    nd = ~done_ref
    assert np.array_equal(got["terminated"], ref["terminated"]), f"{tag}: terminated mask differs"
    assert np.array_equal(got["truncated"], ref["truncated"]), f"{tag}: truncated mask differs"
    assert np.array_equal(got["elapsed"][nd], ref["elapsed"][nd]), f"{tag}: elapsed differs"
    assert np.all(got["elapsed"][done_ref] == 0), f"{tag}: elapsed not zeroed by autoreset"
    pairs = [("obs", got["obs"][nd], ref["obs"][nd]), ("final_obs", got["final_obs"][done_ref], ref["final_obs"][done_ref])]
    for what, g, r in pairs:
        if strict:
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), f"{tag}: {what} not bit-exact"
        else:
            u = ulps32(g, r)
            assert u.size == 0 or u.max() <= MAX_OBS_ULPS, f"{tag}: {what} off by {u.max()} float32 ulps"
            np.testing.assert_allclose(g, r, rtol=OBS_RTOL, atol=1e-30, err_msg=f"{tag}: {what}")
    if strict:
        assert np.array_equal(got["reward"], ref["reward"]), f"{tag}: reward not bit-exact"
        assert np.array_equal(got["state"][nd], ref["state"][nd]), f"{tag}: fp64 state not bit-exact"
    else:
        ra = REWARD_ATOL.get(tag.split()[0], REWARD_ATOL_DEFAULT)
        np.testing.assert_allclose(got["reward"], ref["reward"], rtol=REWARD_RTOL, atol=ra, err_msg=f"{tag}: reward")
        np.testing.assert_allclose(got["state"][nd], ref["state"][nd], rtol=STATE_RTOL, atol=STATE_ATOL,
                                   err_msg=f"{tag}: fp64 state")


def run_p1(engine_cls, name, strict, kind="p1", golden=None):
    """Single raw-env steps from hand-set states (golden <env>_p1.npz; kind="p1_threshold": Acrobot states whose post-step
    height sits within ulps of the termination threshold, make_golden_goal.py; golden=: vectors made on the spot)."""
    g = load_golden(name, kind) if golden is None else golden
    n = len(g["action"])
    eng = engine_cls(name, n, 0, autoreset=False)
    elapsed = np.where(g["fresh"] == 1, 0, 5).astype(np.int32)
    eng.set_state(g["state0"].T, elapsed)
    obs, rew, term, trunc, fin = eng.step(g["action"])
    st, el = eng.get_state()
    done = np.zeros(n, dtype=bool)
    got = dict(obs=obs, reward=rew, terminated=term, truncated=trunc, final_obs=fin, state=st.T, elapsed=el)
    ref = dict(obs=g["obs"], reward=g["reward"], terminated=g["terminated"].astype(bool),
               truncated=np.zeros(n, dtype=bool), final_obs=g["obs"], state=g["state1"], elapsed=elapsed + 1)
    compare_step(f"{name} {kind.upper()}", got, ref, done, strict)
    return int(term.sum())


def run_p1_variants(engine_cls, name, strict):
    """Single raw-env steps with NON-DEFAULT physics attributes (golden <env>_p1_variants.npz, make_golden_variants.py: the
    reference run with the attributes set on the raw env).  The engine gets the same parameter vector through its set_params."""
    g = load_golden(name, "p1_variants")
    V, n = g["action"].shape
    total = 0
    for v in range(V):
        eng = engine_cls(name, n, 0, autoreset=False)
        eng.set_params(g["params"][v])
        elapsed = np.where(g["fresh"][v] == 1, 0, 5).astype(np.int32)
        eng.set_state(g["state0"][v].T, elapsed)
        obs, rew, term, trunc, fin = eng.step(g["action"][v])
        st, el = eng.get_state()
        done = np.zeros(n, dtype=bool)
        got = dict(obs=obs, reward=rew, terminated=term, truncated=trunc, final_obs=fin, state=st.T, elapsed=el)
        ref = dict(obs=g["obs"][v], reward=g["reward"][v], terminated=g["terminated"][v].astype(bool),
                   truncated=np.zeros(n, dtype=bool), final_obs=g["obs"][v], state=g["state1"][v], elapsed=elapsed + 1)
        compare_step(f"{name} P1[variant {v}]", got, ref, done, strict)
        total += int(term.sum())
    return total


def run_p2(engine_cls, name, tag, strict, golden=None):
    """Teacher-forced replay of a SyncVectorEnv trajectory (golden <env>_p2_<tag>.npz): before every step the
    engine is given the reference's pre-step fp64 state and elapsed counters (so PCG64-vs-Philox resets cannot
    desynchronise the two), then one vector step is compared output by output."""
    g = load_golden(name, f"p2_{tag}") if golden is None else golden
    T, N = g["action"].shape
    eng = engine_cls(name, N, int(g["max_episode_steps"]), autoreset=True)
    ndone = 0
    for t in range(T):
        eng.set_state(g["state_pre"][t].T, g["elapsed_pre"][t])
        obs, rew, term, trunc, fin = eng.step(g["action"][t])
        st, el = eng.get_state()
        done = (g["terminated"][t] | g["truncated"][t]).astype(bool)
        assert np.array_equal(done, g["final_mask"][t].astype(bool))
        got = dict(obs=obs, reward=rew, terminated=term, truncated=trunc, final_obs=fin, state=st.T, elapsed=el)
        ref = dict(obs=g["obs"][t], reward=g["reward"][t], terminated=g["terminated"][t].astype(bool),
                   truncated=g["truncated"][t].astype(bool), final_obs=g["final_obs"][t], state=g["state_post"][t],
                   elapsed=g["elapsed_post"][t])
        compare_step(f"{name} P2[{tag}] t={t}", got, ref, done, strict)
        ndone += int(done.sum())
    return ndone


# ---- NormalizeObservation / NormalizeReward -------------------------------------------------------
NORM_CASES = ["CartPole", "Pendulum", "Acrobot", "MountainCarContinuous"]


def load_norm_golden(name):
    return np.load(os.path.join(GOLDEN, f"normalize_{name}.npz"))


def norm_obs_bound(raw_obs, ref_norm, rtol=1e-5):
    """Tolerance of `exact-sum batch moments` against the reference's float32 np.mean / np.var, per element of the
    normalised observations [T][n][O].

    The reference accumulates each batch column in float32: its batch mean carries an absolute error of
    ~sqrt(n) * 2^-24 * max|x| (sequential sum, random-walk growth) and its batch var, formed from float32 (x - mean)^2,
    a relative error of ~2^-23 * max|x| / std.  Propagated through y = (x - mean) / sqrt(var + eps) and through the
    running merge (errors of earlier batches stay in the running statistics) this gives, per element,
        |dy| <= rtol * |y|  +  2^-24 * (2 + sqrt(n)) * max|x_col| / std_col * (1 + |y|)
    with std_col the running std implied by the reference's own output scale; rtol is north_star's fp32 rtol.
    (The worst golden element sits at 0.25 of this bound.)"""
    x = np.asarray(raw_obs, dtype=np.float64)
    T, n, O = x.shape
    xmax = np.abs(x).max(axis=(0, 1))                                  # [O]
    # running std recovered from the reference's outputs: y = (x - mean)/std  =>  std = range(x) / range(y) per batch
    rng_x = x.max(axis=1) - x.min(axis=1)                              # [T][O]
    rng_y = ref_norm.max(axis=1) - ref_norm.min(axis=1)
    std = np.where(rng_y > 0, rng_x / np.where(rng_y > 0, rng_y, 1), np.inf)   # [T][O]
    scale = (2.0 ** -24) * (2.0 + np.sqrt(n)) * xmax[None, :] / std    # [T][O]
    return rtol * np.abs(ref_norm) + scale[:, None, :] * (1.0 + np.abs(ref_norm)) + 1e-300


# ---- tabular toy_text envs ---------------------------------------------------------------------------
TOYTEXT_CASES = ["FrozenLake-v1", "FrozenLake8x8-v1", "FrozenLake-v1_deterministic", "FrozenLake-v1_limit7", "Taxi-v3",
                 "CliffWalking-v0"]


def load_toytext_golden(tag):
    return np.load(os.path.join(GOLDEN, f"toytext_{tag}.npz"))


def toytext_mdp(g):
    """The product's MDP builder for a golden case (gym_amd.toy_text is host-only code: importable without a GPU)."""
    from gym_amd import toy_text

    gid = str(g["id"])
    kw = {} if gid in ("Taxi-v3", "CliffWalking-v0") else {"is_slippery": bool(g["is_slippery"])}
    return toy_text.TOY_TEXT_REGISTRY[gid].build(**kw)


def replay_toytext(g, make_engine):
    """Feed the golden's actions and the reference's recorded uniforms to an engine and compare every output bit for bit.
    make_engine(mdp, n, limit) -> object with set_state(state, elapsed) and step(actions, uniforms) -> dict as OracleTabEnv."""
    mdp = toytext_mdp(g)
    T, n = g["actions"].shape
    eng = make_engine(mdp, n, int(g["max_episode_steps"]))
    eng.set_state(g["obs0"].astype(np.int32), np.zeros(n, np.int32))
    ndone = 0
    for t in range(T):
        out = eng.step(g["actions"][t], g["uniforms"][t])
        done = g["final_mask"][t]
        assert np.array_equal(out["obs"], g["obs"][t]), t
        assert np.array_equal(out["reward"], g["reward"][t]), t
        assert np.array_equal(out["terminated"], g["terminated"][t]) and np.array_equal(out["truncated"], g["truncated"][t]), t
        prob = out["prob"].astype(np.int64) if g["prob_is_int"][t] else out["prob"]   # VectorEnv._add_info's dtype quirk
        assert np.array_equal(prob, g["prob"][t]), t
        assert np.array_equal(out["terminated"] | out["truncated"], done), t
        assert np.array_equal(out["final_obs"][done], g["final_obs"][t][done]), t
        assert np.array_equal(out["final_prob"][done], g["final_prob"][t][done]), t
        if mdp.action_mask is not None:
            assert np.array_equal(mdp.action_mask[out["obs"]], g["step_action_mask"][t]), t
        ndone += int(done.sum())
    return ndone


# ---- Blackjack-v1 -------------------------------------------------------------------------------------------------------
BLACKJACK_CASES = ["sab", "natural", "plain"]


def replay_blackjack(tag, make_engine):
    """Deal an engine the cards the reference drew (tests/golden/blackjack_<tag>.npz) and compare every output exactly.
    make_engine(n, natural, sab) -> object with reset(cards[n,4]) -> obs[3,n] and step(actions, cards[n,24]) -> dict."""
    g = np.load(os.path.join(GOLDEN, f"blackjack_{tag}.npz"))
    T, n = g["actions"].shape
    eng = make_engine(n, bool(g["natural"]), bool(g["sab"]))
    assert np.array_equal(eng.reset(g["cards0"]), g["obs0"])
    ndone = 0
    for t in range(T):
        out = eng.step(g["actions"][t], g["cards"][t])
        done = g["final_mask"][t]
        assert np.array_equal(out["obs"], g["obs"][t]), t
        assert np.array_equal(out["reward"], g["reward"][t]), t
        assert np.array_equal(out["terminated"], g["terminated"][t]) and np.array_equal(out["truncated"], g["truncated"][t]), t
        assert np.array_equal(out["final_obs"][:, done], g["final_obs"][t][:, done]), t
        ndone += int(done.sum())
    return ndone, g


def run_p1_nonfinite(engine_cls, name, strict):
    """<env>_p1_nonfinite.npz (tests/golden/make_golden_nonfinite.py): NaN / +-Inf Box actions and state components through one step.
    NaNs must sit exactly where the reference's do (no clamp may swallow one, none may appear); everything else to the usual bars:
    bit-exact for the oracle (strict), float32-ulp / rtol bars for the engine."""
    g = load_golden(name, "p1_nonfinite")
    n = len(g["action"])
    eng = engine_cls(name, n, 0, autoreset=False)
    elapsed = np.where(g["fresh"] == 1, 0, 5).astype(np.int32)
    eng.set_state(g["state0"].T, elapsed)
    with np.errstate(all="ignore"):
        obs, rew, term, trunc, fin = eng.step(g["action"])
    st = eng.get_state()[0].T
    assert np.array_equal(term, g["terminated"].astype(bool)), f"{name}: terminated mask differs on non-finite inputs"
    assert not trunc.any()
    for what, got, ref in (("obs", obs, g["obs"]), ("reward", rew, g["reward"]), ("state", st, g["state1"])):
        assert np.array_equal(np.isnan(got), np.isnan(ref)), \
            f"{name}: {what}: NaN where the reference has none, or a NaN swallowed, in rows {np.flatnonzero((np.isnan(got) != np.isnan(ref)).reshape(n, -1).any(axis=1))[:8]}"
        ok = ~np.isnan(ref)
        assert np.array_equal(np.isinf(got[ok]), np.isinf(ref[ok])) and np.array_equal(np.sign(got[ok][np.isinf(ref[ok])]), np.sign(ref[ok][np.isinf(ref[ok])]))
        fin_ = ok & ~np.isinf(ref)
        if strict:
            assert np.array_equal(got[fin_], ref[fin_]), f"{name}: {what} not bit-exact"
        elif what == "obs":
            assert ulps32(got[fin_], ref[fin_]).max() <= MAX_OBS_ULPS
        else:
            atol = REWARD_ATOL.get(name, REWARD_ATOL_DEFAULT) if what == "reward" else STATE_ATOL
            np.testing.assert_allclose(got[fin_], ref[fin_], rtol=REWARD_RTOL if what == "reward" else STATE_RTOL, atol=atol)
    return int(np.isnan(g["state1"]).any(axis=1).sum())


def run_cartpole_beyond(engine_cls, strict):
    """CartPole_p2_beyond.npz (tests/golden/make_golden_cartpole_beyond.py): envs stepped on after they terminated, no reset in between
    (MXV_FLAG_NO_AUTORESET): the fall pays 1.0, every later terminated step 0.0 (cartpole.py:169-184)."""
    g = load_golden("CartPole", "p2_beyond")
    T, n = g["action"].shape
    eng = engine_cls("CartPole", n, 0, autoreset=False)
    eng.set_state(g["state0"].T, np.full(n, 3, np.int32))
    zero = 0
    for t in range(T):
        obs, rew, term, trunc, fin = eng.step(g["action"][t])
        assert np.array_equal(term, g["terminated"][t].astype(bool)), f"step {t}: terminated"
        assert np.array_equal(rew, g["reward"][t]), f"step {t}: rewards {rew[rew != g['reward'][t]][:4]} vs {g['reward'][t][rew != g['reward'][t]][:4]}"
        st = eng.get_state()[0].T
        if strict:
            assert np.array_equal(obs, g["obs"][t]) and np.array_equal(st, g["state_post"][t])
        else:
            assert ulps32(obs, g["obs"][t]).max() <= MAX_OBS_ULPS
            np.testing.assert_allclose(st, g["state_post"][t], rtol=1e-11, atol=1e-12)
        zero += int((rew == 0).sum())
    assert zero > 100
    # a reset (here: a fresh state) clears the mark: the next fall pays 1.0 again
    eng.set_state(g["state0"].T, np.full(n, 3, np.int32))
    obs, rew, term, trunc, fin = eng.step(g["action"][0])
    assert np.array_equal(rew, g["reward"][0]) and np.all(rew == 1.0)
    return zero


# ---- Blackjack: exact probabilities of a deck of iid draws (deck = [1..10, 10, 10, 10], gym/envs/toy_text/blackjack.py:14-19) -------------
DECK_P = np.array([1.0 / 13] * 9 + [4.0 / 13])       # P(card = 1..10)


def chi2_ok(counts, probs, sigmas=5.0):
    """Pearson chi-square of observed counts against probabilities, accepted within `sigmas` standard deviations of its mean (df)."""
    counts, probs = np.asarray(counts, dtype=np.float64).ravel(), np.asarray(probs, dtype=np.float64).ravel()
    exp = counts.sum() * probs
    chi2 = float(((counts - exp) ** 2 / exp).sum())
    df = counts.size - 1
    return chi2 < df + sigmas * np.sqrt(2.0 * df), chi2


def dealer_score_distribution(raw_sum, ace):
    """P(final dealer score) for a dealer that starts at (raw sum with aces as 1, holds an ace) and draws iid deck cards until its total
    (one ace as 11 when that does not bust, blackjack.py:26-33) reaches 17 (:132): dict score -> probability, score 0 = bust (:36-41)."""
    from functools import lru_cache

    @lru_cache(maxsize=None)
    def go(s, a):
        total = s + 10 if (a and s + 10 <= 21) else s
        if total >= 17:
            return ((0 if total > 21 else total, 1.0),)
        out = {}
        for c in range(1, 11):
            for sc, pr in go(s + c, a or c == 1):
                out[sc] = out.get(sc, 0.0) + pr * DECK_P[c - 1]
        return tuple(sorted(out.items()))

    return dict(go(int(raw_sum), bool(ace)))


def reference_wrapper_stub(name):
    """A stand-in for the reference's `gym.wrappers.<name>` class where the reference is not importable (the GPU box): gym_amd.make
    recognises per-sub-env wrappers by class name AND home module (gym.wrappers.* or gym_amd.*) and never calls them."""
    snake = "".join("_" + c.lower() if c.isupper() else c for c in name).lstrip("_")
    return type(name, (), {"__module__": f"gym.wrappers.{snake}"})


# ---- gym.vector.make(..., wrappers=[TimeLimit, NormalizeObservation, NormalizeReward, RecordEpisodeStatistics]) replays ---------------------
def replay_vector_make_normalize(name, exact, device=None):
    """tests/golden/vector_make_normalize_<name>.npz (made by tests/golden/make_golden_vector_make.py: THE REFERENCE with those four wrappers
    around every sub-env) replayed through gym_amd's per-sub-env wrappers over whatever engine gym_amd.make builds (the HIP engine on the GPU
    box, the oracle-backed handle in the CPU tests): teacher-forced pre-step states, the reference's own reset observations injected for
    the finished sub-envs (its resets are PCG64 draws), then every output compared — batched float32 observations, float64 final
    observations, normalised float64 rewards, episode returns of NORMALISED rewards.  exact: bit-equal (oracle) / within subnorm_bounds — the
    engine's raw-output tolerances PROPAGATED through the statistics, derived below, no hand-picked number.  device: None = the wrappers'
    own choice (host NumPy below SUBENV_DEVICE_MIN sub-envs), True = the mxv_subnorm_* kernels."""
    import functools

    import gym_amd
    from gym_amd.wrappers import (RecordEpisodeStatistics, SubEnvEpisodeStatistics, SubEnvNormalizeObservation, SubEnvNormalizeReward,
                                  _VectorWrapper)

    TimeLimit, NormalizeObservation, NormalizeReward = (reference_wrapper_stub(n) for n in ("TimeLimit", "NormalizeObservation", "NormalizeReward"))

    g = np.load(os.path.join(GOLDEN, f"vector_make_normalize_{name}.npz"))
    T, N = g["terminated"].shape
    K, gamma = int(g["max_episode_steps"]), float(g["gamma"])
    # (1) the mapping: the chain gym_amd.make builds from the reference-style wrappers list
    built = gym_amd.make(GYM_IDS[name], num_envs=N, wrappers=[functools.partial(TimeLimit, max_episode_steps=K), NormalizeObservation,
                                                              functools.partial(NormalizeReward, gamma=gamma), RecordEpisodeStatistics])
    chain, e = [], built
    while isinstance(e, _VectorWrapper):
        chain.append(type(e).__name__)
        e = e.env
    assert chain == ["SubEnvEpisodeStatistics", "RecordEpisodeStatistics", "SubEnvNormalizeReward", "SubEnvNormalizeObservation"], chain
    assert built.get_attr("_max_episode_steps") == (K,) * N and built.env.env.gamma == gamma
    built.close()

    # (2) the numbers: the same chain with a shim above the engine that hands the finished sub-envs the reference's reset observations
    class ReferenceResets(_VectorWrapper):
        t = 0

        def reset(self, **kw):
            obs, infos = self.env.reset(**kw)
            return g["raw_obs0"].copy(), infos

        def step(self, action):
            obs, rew, term, trunc, infos = self.env.step(action)
            done = term | trunc
            obs = obs.copy()
            obs[done] = g["raw_obs_post"][self.t][done]
            return obs, rew, term, trunc, infos

    base = gym_amd.make(GYM_IDS[name], num_envs=N, max_episode_steps=K)
    shim = ReferenceResets(base)
    env = SubEnvEpisodeStatistics(SubEnvNormalizeReward(SubEnvNormalizeObservation(shim, device=device), gamma=gamma, device=device))
    bound = None if exact else subnorm_bounds(g, name)
    obs0, _ = env.reset(seed=1)
    assert obs0.dtype == np.float32 and np.array_equal(obs0, g["obs0"])
    episodes = 0
    for t in range(T):
        shim.t = t
        base.handle.set_state(np.ascontiguousarray(g["state_pre"][t].T), g["elapsed_pre"][t])
        obs, rew, term, trunc, infos = env.step(g["action"][t])
        done = g["terminated"][t] | g["truncated"][t]
        assert np.array_equal(term, g["terminated"][t]) and np.array_equal(trunc, g["truncated"][t]), t
        assert obs.dtype == np.float32 and rew.dtype == np.float64
        if exact:
            assert np.array_equal(obs, g["obs"][t]) and np.array_equal(rew, g["reward"][t]), t
        else:
            assert (np.abs(obs - g["obs"][t]) <= bound["obs"][t]).all(), (t, np.abs(obs - g["obs"][t]).max(), bound["obs"][t].min())
            assert (np.abs(rew - g["reward"][t]) <= bound["reward"][t]).all(), (t, np.abs(rew - g["reward"][t]).max())
        for i in np.flatnonzero(done):
            fo, ep = infos["final_observation"][i], infos["final_info"][i]["episode"]
            assert fo.dtype == np.float64 and isinstance(ep["r"], np.float32) and ep["l"] == g["ep_l"][t][i]
            if exact:
                assert np.array_equal(fo, g["final_obs"][t][i]) and ep["r"] == g["ep_r"][t][i], (t, i)
            else:
                assert (np.abs(fo - g["final_obs"][t][i]) <= bound["final_obs"][t][i]).all(), (t, i)
                assert abs(float(ep["r"]) - float(g["ep_r"][t][i])) <= bound["ep_r"][t][i], (t, i)
            episodes += 1
    assert episodes == int((g["terminated"] | g["truncated"]).sum()) > 50
    env.close()
    return episodes


def replay_vector_make_transform(exact, device=None):
    """tests/golden/vector_make_transform_Pendulum.npz (THE REFERENCE: wrappers=[TimeLimit(12), ClipAction, NormalizeObservation,
    TransformObservation(clip), NormalizeReward, TransformReward(clip)] around every sub-env — the continuous-control PPO recipe) through
    the chain gym_amd.make builds from the same list, teacher-forced like replay_vector_make_normalize.  exact: bit-equal (oracle-backed
    handle); else the engine's raw tolerances propagated through the statistics (subnorm_bounds; a clip only shrinks a difference)."""
    import functools

    import gym_amd
    from gym_amd.wrappers import _VectorWrapper

    g = np.load(os.path.join(GOLDEN, "vector_make_transform_Pendulum.npz"))
    T, N = g["terminated"].shape
    K, gamma, oc, rc = int(g["max_episode_steps"]), float(g["gamma"]), float(g["obs_clip"]), float(g["reward_clip"])
    W = {n: reference_wrapper_stub(n) for n in ("TimeLimit", "ClipAction", "NormalizeObservation", "TransformObservation", "NormalizeReward", "TransformReward")}
    wrappers = [functools.partial(W["TimeLimit"], max_episode_steps=K), W["ClipAction"], W["NormalizeObservation"],
                functools.partial(W["TransformObservation"], f=lambda o: np.clip(o, -oc, oc)),
                functools.partial(W["NormalizeReward"], gamma=gamma), functools.partial(W["TransformReward"], f=lambda r: np.clip(r, -rc, rc))]
    env = gym_amd.make("Pendulum-v1", num_envs=N, wrappers=wrappers)
    chain, e = [], env
    while isinstance(e, _VectorWrapper):
        chain.append(e)
        e = e.env
    assert [type(w).__name__ for w in chain] == ["SubEnvTransformReward", "SubEnvNormalizeReward", "SubEnvTransformObservation", "SubEnvNormalizeObservation"]
    base, norm_obs = e, chain[-1]

    # the reference's reset observations for the finished sub-envs (its resets are PCG64 draws): a shim right above the engine
    class ReferenceResets(_VectorWrapper):
        t = 0

        def reset(self, **kw):
            obs, infos = self.env.reset(**kw)
            return g["raw_obs0"].copy(), infos

        def step(self, action):
            obs, rew, term, trunc, infos = self.env.step(action)
            done = term | trunc
            obs = obs.copy()
            obs[done] = g["raw_obs_post"][self.t][done]
            return obs, rew, term, trunc, infos

    if device is None:          # the chain as make() built it (eight sub-envs: the wrappers' NumPy form), the shim slipped in underneath
        shim = ReferenceResets(base)
        norm_obs.env = shim
    else:                       # the same chain by hand with the device / NumPy form of both normalisers forced
        from gym_amd.wrappers import SubEnvNormalizeObservation, SubEnvNormalizeReward, SubEnvTransformObservation, SubEnvTransformReward

        env.close()
        base = gym_amd.make("Pendulum-v1", num_envs=N, max_episode_steps=K)
        shim = ReferenceResets(base)
        norm_obs = SubEnvNormalizeObservation(shim, device=device)
        env = SubEnvTransformReward(SubEnvNormalizeReward(SubEnvTransformObservation(norm_obs, lambda o: np.clip(o, -oc, oc)), gamma=gamma, device=device),
                                    lambda r: np.clip(r, -rc, rc))
    assert norm_obs._wide is True
    bound = None if exact else subnorm_bounds(g, "Pendulum")
    obs0, _ = env.reset(seed=1)
    assert obs0.dtype == np.float32 and np.array_equal(obs0, g["obs0"])
    episodes = 0
    for t in range(T):
        shim.t = t
        base.handle.set_state(np.ascontiguousarray(g["state_pre"][t].T), g["elapsed_pre"][t])
        obs, rew, term, trunc, infos = env.step(g["action"][t])
        done = g["terminated"][t] | g["truncated"][t]
        assert np.array_equal(term, g["terminated"][t]) and np.array_equal(trunc, g["truncated"][t]), t
        assert obs.dtype == np.float32 and rew.dtype == np.float64 and np.abs(obs).max() <= oc and np.abs(rew).max() <= rc
        if exact:
            assert np.array_equal(obs, g["obs"][t]) and np.array_equal(rew, g["reward"][t]), t
        else:
            assert (np.abs(obs - g["obs"][t]) <= bound["obs"][t]).all() and (np.abs(rew - g["reward"][t]) <= bound["reward"][t]).all(), t
        for i in np.flatnonzero(done):
            fo = infos["final_observation"][i]
            assert fo.dtype == np.float64 and np.abs(fo).max() <= oc
            if exact:
                assert np.array_equal(fo, g["final_obs"][t][i]), (t, i)
            else:
                assert (np.abs(fo - g["final_obs"][t][i]) <= bound["final_obs"][t][i]).all(), (t, i)
            episodes += 1
    assert episodes == int((g["terminated"] | g["truncated"]).sum()) > 40
    assert chain[0]._map.batched is True or device is not None
    env.close()
    return episodes


def subnorm_bounds(g, name):
    """Per-element tolerances of the per-sub-env Normalize* replay on the device, DERIVED from the engine's raw-output bars
    (MAX_OBS_ULPS float32 ulps on observations; REWARD_RTOL / REWARD_ATOL on rewards) by first-order propagation through the statistics.
    u = 2^-23.  With X the column's largest |raw observation| in the run and s = sqrt(var + eps) the reference's own running std of
    that sub-env and column at that update (recomputed here with the wrappers' arithmetic from the reference's raw rows):
        |dx| <= MAX_OBS_ULPS u X;   the running mean is a convex combination of past rows (and the prior 0):  |dm| <= |dx|
        var = E_w[x^2] - m^2 over the same weights:  |dvar| <= 2 X |dx| + 2 X |dm| = 4 X |dx|,   |ds| <= |dvar| / (2 s)
        y = (x - m) / s:   |dy| <= (|dx| + |dm|) / s + |y| |ds| / s  =  2 |dx| / s * (1 + |y| X / s),   + u |y| for the float32 store
    Rewards (a = REWARD_ATOL, rho = REWARD_RTOL, G = 1 / (1 - gamma), Rmax = the largest |discounted return| in the run):
        |dr| <= a + rho |r|;   |dR| <= G max|dr| =: D;   |dm| <= D;   |dvar| <= 4 Rmax D;   |ds| <= 2 Rmax D / s
        out = r / s:   |dout| <= |dr| / s + |out| 2 Rmax D / s^2,   + 2^-50 |out| for the fp64 operations themselves
    Episode returns (float32 running sums of <= K normalised rewards): sum of the |dout| bounds + K 2^-24 sum |out|."""
    from gym_amd.wrappers import _PerEnvMeanStd
    from oracle.oracle import OracleVecEnv

    T, N = g["terminated"].shape
    O = g["obs"].shape[2]
    K, gamma, eps = int(g["max_episode_steps"]), float(g["gamma"]), 1e-8
    u = 2.0 ** -23
    env = OracleVecEnv(ENV_IDS[name], N, K, seed=1)
    env.reset(seed=1)
    orm, rrm, ret = _PerEnvMeanStd(N, (O,)), _PerEnvMeanStd(N, ()), np.zeros(N)
    orm.update(g["raw_obs0"])
    raws, fins, rews, s_y, s_f, s_r, rets = [], [], [], [], [], [], []
    for t in range(T):
        env.state[:] = g["state_pre"][t].T
        env.elapsed[:] = g["elapsed_pre"][t]
        _, rew, term, trunc, fin, _ = env.step(g["action"][t])
        done = term | trunc
        idx = np.flatnonzero(done)
        first = g["raw_obs_post"][t].copy()
        first[idx] = fin[idx]
        orm.update(first)
        s_f.append(np.sqrt(orm.var + eps))
        orm.update(g["raw_obs_post"][t][idx], idx)
        s_y.append(np.sqrt(orm.var + eps))
        ret = ret * gamma + rew
        rets.append(np.abs(ret))
        rrm.update(ret)
        s_r.append(np.sqrt(rrm.var + eps))
        ret[done] = 0.0
        raws.append(np.abs(first)), raws.append(np.abs(g["raw_obs_post"][t])), fins.append(fin), rews.append(rew)
    X = np.max(np.stack(raws), axis=(0, 1))                      # [O]
    dx = MAX_OBS_ULPS * u * X
    s_y, s_f, s_r, rews = np.stack(s_y), np.stack(s_f), np.stack(s_r), np.stack(rews)
    y, yf = np.abs(g["obs"].astype(np.float64)), np.abs(np.nan_to_num(g["final_obs"]))
    obs_b = u * y + 2 * dx / s_y * (1 + y * X / s_y)
    fin_b = 2.0 ** -50 * yf + 2 * dx / s_f * (1 + yf * X / s_f)
    a, G, Rmax = REWARD_ATOL.get(name, REWARD_ATOL_DEFAULT), 1.0 / (1.0 - gamma), float(np.max(rets))
    dr = a + REWARD_RTOL * np.abs(rews)
    D = G * float(dr.max())
    out = np.abs(g["reward"])
    rew_b = 2.0 ** -50 * out + dr / s_r + out * 2 * Rmax * D / s_r ** 2
    ep_b, acc_b, acc_o = np.zeros((T, N)), np.zeros(N), np.zeros(N)
    for t in range(T):
        acc_b, acc_o = acc_b + rew_b[t], acc_o + out[t]
        ep_b[t] = acc_b + K * 2.0 ** -24 * acc_o
        done = g["terminated"][t] | g["truncated"][t]
        acc_b, acc_o = np.where(done, 0.0, acc_b), np.where(done, 0.0, acc_o)
    return dict(obs=obs_b, final_obs=fin_b, reward=rew_b, ep_r=ep_b)


def replay_vector_make_clipaction(exact):
    """tests/golden/vector_make_clipaction_MountainCarContinuous.npz (THE REFERENCE: gym.vector.make("MountainCarContinuous-v0", 6,
    wrappers=ClipAction) stepped with actions far outside [-1, 1]) through gym_amd.make(..., wrappers=ClipAction): the reward's penalty
    must be charged on the CLIPPED action (continuous_mountain_car.py:169 under clip_action.py:33-43) — ADVICE r5: it used to be treated as
    an identity and charged -2.5 where the reference charges -0.1."""
    import gym_amd

    g = np.load(os.path.join(GOLDEN, "vector_make_clipaction_MountainCarContinuous.npz"))
    T, N = g["terminated"].shape
    env = gym_amd.make("MountainCarContinuous-v0", num_envs=N, wrappers=reference_wrapper_stub("ClipAction"))
    base = env.unwrapped
    env.reset(seed=1)
    elapsed = np.zeros(N, np.int32)        # steps since each sub-env's last reset (its first step afterwards computes in float64: App. A.5)
    for t in range(T):
        base.handle.set_state(np.ascontiguousarray(g["state_pre"][t].T), elapsed)
        obs, rew, term, trunc, _ = env.step(g["action"][t])
        elapsed = np.where(g["terminated"][t] | g["truncated"][t], 0, elapsed + 1).astype(np.int32)
        assert np.array_equal(term, g["terminated"][t]) and np.array_equal(trunc, g["truncated"][t]), t
        assert rew.min() >= -0.1 - 1e-12
        if exact:
            assert np.array_equal(rew, g["reward"][t]) and np.array_equal(obs, g["obs"][t]), t
        else:
            np.testing.assert_allclose(rew, g["reward"][t], rtol=REWARD_RTOL, atol=1e-300, err_msg=f"t={t}")
            assert ulps32(obs, g["obs"][t]).max() <= MAX_OBS_ULPS, t
    env.close()
    return T


def replay_vector_make_rescaleaction(name, exact):
    """tests/golden/vector_make_rescaleaction_<name>.npz (THE REFERENCE: gym.vector.make(id, 6, wrappers=partial(RescaleAction, min_action=a,
    max_action=b)) stepped with actions in [a, b]) through gym_amd.make(..., wrappers=partial(RescaleAction, ...)): the affine map onto the
    sub-env's own bounds and the clip (rescale_action.py:64-83), then the step."""
    import functools

    import gym_amd

    g = np.load(os.path.join(GOLDEN, f"vector_make_rescaleaction_{name}.npz"))
    T, N = g["terminated"].shape
    gid = {"Pendulum": "Pendulum-v1", "MountainCarContinuous": "MountainCarContinuous-v0"}[name]
    a, b = float(g["min_action"]), float(g["max_action"])
    env = gym_amd.make(gid, num_envs=N, wrappers=functools.partial(reference_wrapper_stub("RescaleAction"), min_action=a, max_action=b))
    assert env.single_action_space.low[0] == a and env.single_action_space.high[0] == b and env.action_space.shape == (N, 1)
    base = env.unwrapped
    env.reset(seed=1)
    elapsed = np.zeros(N, np.int32)
    for t in range(T):
        base.handle.set_state(np.ascontiguousarray(g["state_pre"][t].T), elapsed)
        obs, rew, term, trunc, _ = env.step(g["action"][t])
        elapsed = np.where(g["terminated"][t] | g["truncated"][t], 0, elapsed + 1).astype(np.int32)
        assert np.array_equal(term, g["terminated"][t]) and np.array_equal(trunc, g["truncated"][t]), t
        if exact:
            assert np.array_equal(rew, g["reward"][t]) and np.array_equal(obs, g["obs"][t]), t
        else:
            np.testing.assert_allclose(rew, g["reward"][t], rtol=REWARD_RTOL, atol=REWARD_ATOL.get(name, REWARD_ATOL_DEFAULT), err_msg=f"t={t}")
            assert ulps32(obs, g["obs"][t]).max() <= MAX_OBS_ULPS, t
    with pytest.raises(AssertionError):            # outside [a, b]: the reference's own assertion (rescale_action.py:73-77)
        env.step(np.full((N, 1), b + 1.0, np.float32))
    env.close()
    return T


# ---- episode statistics over the toy_text engines (tests/golden/toytext_stats_*.npz: the reference's RecordEpisodeStatistics) --------------
TOYTEXT_STATS_CASES = ["FrozenLake-v1", "Taxi-v3", "Blackjack-v1"]


def replay_toytext_stats(tag, make_engine):
    """The reference's own run of RecordEpisodeStatistics over gym.vector.make(<tag>, 8) — vector-level wrapper and per-sub-env wrapper,
    the same trajectory (tests/golden/make_golden_toytext_stats.py) — replayed with its recorded uniforms / cards:
    make_engine(g, n) -> object with step(t) -> (reward, terminated, truncated, ep_return float32 [n], ep_length int32 [n]) where the last
    two are valid where terminated | truncated.  Rewards, flags, episode returns and lengths must match bit for bit, in both of the
    reference's dtypes (float64 arrays of the vector-level wrapper; float32 / int32 scalars of the per-sub-env one)."""
    g = np.load(os.path.join(GOLDEN, f"toytext_stats_{tag}.npz"))
    T, n = g["actions"].shape
    eng = make_engine(g, n)
    episodes = 0
    for t in range(T):
        rew, term, trunc, er, el = eng.step(t)
        done = g["ep_mask"][t]
        assert np.array_equal(rew, g["reward"][t]) and np.array_equal(term, g["terminated"][t]) and np.array_equal(trunc, g["truncated"][t]), t
        assert er.dtype == np.float32 and el.dtype == np.int32
        assert np.array_equal(er[done].astype(np.float64), g["ep_r"][t][done]) and np.array_equal(el[done].astype(np.float64), g["ep_l"][t][done]), t
        assert np.array_equal(er[done], g["sub_ep_r"][t][done]) and np.array_equal(el[done], g["sub_ep_l"][t][done]), t
        episodes += int(done.sum())
    assert episodes == int(g["ep_mask"].sum()) > 10
    return episodes, g


def toytext_stats_start(g, mdp=None):
    """What the recorded reset left the sub-envs in: tabular — categorical_sample(initial_state_distrib, u) (toy_text/utils.py:4-8) of the
    recorded uniform; Blackjack — the four recorded cards (dealer's hand first, blackjack.py:157-158)."""
    if mdp is None:
        return g["first"].astype(np.int8)
    u = g["first"][:, 0]
    return np.array([int(np.argmax(mdp.initial_cum > x)) for x in u], dtype=np.int32)
