"""Parity of the HIP engine, called through the C ABI, against (a) the reference's golden vectors,
(b) SURVEY App. B known answers, (c) the oracle on seeded random rollouts incl. Philox actions + autoreset,
(d) size-independent properties at BASELINE.json's full sizes.

Bars: terminated / truncated / elapsed / sampled discrete actions / reset states: bit-exact.
Observations: north_star's fp32 rtol=1e-5 and additionally <= MAX_OBS_ULPS float32 ulps; fp64 state and reward
within helpers.STATE_RTOL / REWARD_RTOL (device sin/cos and x*x may differ from glibc in the last fp64 bit)."""
import numpy as np
import pytest

from helpers import (DISCRETE, ENV_IDS, ENV_NAMES, GYM_IDS, LIMITS, MAX_OBS_ULPS, OBS_RTOL, REWARD_ATOL, REWARD_ATOL_DEFAULT, REWARD_RTOL,
                     HipEngine, OracleEngine, run_cartpole_beyond, run_p1, run_p1_nonfinite, run_p2, ulps32)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ENV_NAMES)
def test_golden_single_steps(name):
    run_p1(HipEngine, name, strict=False)


@pytest.mark.parametrize("tag", ["default", "short"])
@pytest.mark.parametrize("name", ENV_NAMES)
def test_golden_trajectories(name, tag):
    run_p2(HipEngine, name, tag, strict=False)


@pytest.mark.parametrize("name", ["Acrobot", "MountainCar", "MountainCarContinuous"])
def test_golden_goal_reaching_trajectories(name):
    """Terminations (and the autoresets after them) INSIDE trajectories of the envs a random policy never finishes
    (tests/golden/make_golden_goal.py: the reference's SyncVectorEnv under a scripted goal-reaching policy)."""
    assert run_p2(HipEngine, name, "goal", strict=False) >= 8


def test_acrobot_termination_threshold_states():
    """Acrobot_p1_threshold.npz: 4096 single steps whose post-step height -cos(t1) - cos(t2 + t1) sits between 0 and ~15 000 ulps
    of 1.0 from the termination threshold (acrobot.py:235), on both sides, found by bisection on the reference.

    The mask there is decided by the last bit of sin / cos, and the engine's hot path (1.5-ulp sincos, cosines rebuilt by angle
    addition) is not the reference's libm bit for bit.  So inside 2^-40 of the threshold the engine takes the step again as the
    reference writes it, on CORRECTLY ROUNDED sin / cos (gym_amd/csrc/mxv_exact.hpp).  The bar, with no tolerance anywhere:
      * every mask equals the reference's run on a correctly rounded libm (Acrobot_p1_threshold_cr.npz: the reference's own code with
        mpmath-rounded sin / cos, tests/golden/make_golden_acrobot_cr.py);
      * it differs from the reference's glibc run ONLY where that run differs from the correctly rounded one itself — 2 of 4096
        states, both with a height that rounds to exactly 1.0, where glibc's cos returns a neighbour of the rounded value (it does for
        ~0.1 % of arguments) — so the reference's mask there is an accident of its libm build, not of its arithmetic;
      * inside the band the post-step angles and the observations are the correctly-rounded run's bit for bit."""
    from helpers import load_golden

    g = load_golden("Acrobot", "p1_threshold")
    cr = load_golden("Acrobot", "p1_threshold_cr")
    n = len(g["action"])
    eng = HipEngine("Acrobot", n, 0, autoreset=False)
    eng.set_state(g["state0"].T, np.full(n, 5, np.int32))
    obs, rew, term, trunc, fin = eng.step(g["action"])
    want, want_cr = g["terminated"].astype(bool), cr["terminated"].astype(bool)
    ulp = np.abs(g["margin"]) / 2.0 ** -52
    assert (ulp <= 8).sum() > 1000 and (ulp <= 1).sum() > 300
    bad_cr = term != want_cr
    assert not bad_cr.any(), f"{bad_cr.sum()} masks differ from the reference on a correctly rounded libm, {np.sort(ulp[bad_cr])[:10]} ulps from the threshold"
    bad = term != want
    assert np.array_equal(bad, want_cr != want) and bad.sum() <= 4, f"{bad.sum()} masks differ from the glibc run of the reference"
    assert np.all(g["margin"][bad] == 0.0)
    assert np.array_equal(rew, np.where(term, 0.0, -1.0))
    st = eng.get_state()[0].T
    band = np.abs(g["margin"]) < 2.0 ** -41          # well inside the 2^-40 band of the engine's own height
    assert band.sum() > 3000
    assert np.array_equal(st[band, :2], cr["state1"][band, :2]), "post-step angles inside the band: bit for bit"
    assert np.array_equal(obs[band], cr["obs"][band])
    np.testing.assert_allclose(st[band], cr["state1"][band], rtol=3e-16, atol=0)      # velocities: x * x vs the reference's libm pow(x, 2)
    assert ulps32(obs, g["obs"]).max() <= MAX_OBS_ULPS
    np.testing.assert_allclose(st, g["state1"], rtol=1e-12, atol=1e-13)
    assert 1000 < term.sum() < 3000
    print(f"threshold masks: {int(bad.sum())} of {n} differ from the glibc run (both at 0 ulps), 0 from the correctly rounded run")


def test_acrobot_exact_band_inside_a_fused_rollout():
    """The same states through the fused K-step kernel (rollout_kernel_v3, the guarded instantiation after a state injection): the first
    step of a rollout_tape launch must produce the masks of the single-step kernel — the exact path is part of Env<ACROBOT>::step,
    whichever kernel inlines it."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from helpers import load_golden

    g = load_golden("Acrobot", "p1_threshold")
    cr = load_golden("Acrobot", "p1_threshold_cr")
    n = len(g["action"])
    r = DeviceRollout("Acrobot-v1", n, seed=0, action_seed=1)
    r.reset(seed=0)
    r.handle.set_state(np.ascontiguousarray(g["state0"].T), np.full(n, 5, np.int32))
    tape = torch.from_numpy(np.stack([g["action"], np.ones(n, np.int64)])).to(r.device)
    out = r.rollout_tape(tape)
    r.synchronize()
    term = out["terminated"][0].cpu().numpy().astype(bool)
    assert np.array_equal(term, cr["terminated"].astype(bool))
    assert np.array_equal(out["reward"][0].cpu().numpy(), np.where(term, 0.0, -1.0))
    r.close()


@pytest.mark.parametrize("mode", ["broadcast", "per_env"])
def test_acrobot_exact_band_with_runtime_parameters(mode):
    """The exact path takes the launch's parameter values (Env<ACROBOT>::acrobot_exact builds its vector from Par<DEF>): the threshold
    states through the runtime-parameter kernels — one common value set, and a per-env table — with LINK_LENGTH_2 changed, an attribute
    the dynamics never read (acrobot.py:237-277 uses l1 only): the masks must be those of the default-attribute run."""
    from helpers import load_golden

    g = load_golden("Acrobot", "p1_threshold")
    cr = load_golden("Acrobot", "p1_threshold_cr")
    n = len(g["action"])
    eng = HipEngine("Acrobot", n, 0, autoreset=False)
    p = eng.h.get_params()
    if mode == "broadcast":
        p[2] = 1.5
        eng.h.set_params(p)
    else:
        table = np.repeat(p[:, None], n, axis=1)
        table[2] = np.linspace(0.5, 2.0, n)
        eng.h.set_params_per_env(table)
    eng.set_state(g["state0"].T, np.full(n, 5, np.int32))
    obs, rew, term, trunc, fin = eng.step(g["action"])
    assert eng.h.last_launch()["param_mode"] == (0 if mode == "broadcast" else 2)
    assert np.array_equal(term, cr["terminated"].astype(bool))
    band = np.abs(g["margin"]) < 2.0 ** -41
    assert np.array_equal(obs[band], cr["obs"][band])


@pytest.mark.parametrize("name", ENV_NAMES)
def test_non_finite_inputs_pass_through_like_the_reference(name):
    """A NaN Box action (a diverged policy) or state component comes out of the reference as NaN observations / rewards — np.clip and
    Python's comparisons propagate it — and must not be turned into a bound by the device's v_max / v_min clamps."""
    assert run_p1_nonfinite(HipEngine, name, strict=False) > 100


@pytest.mark.parametrize("name", ["Pendulum", "MountainCarContinuous"])
def test_non_finite_actions_inside_a_fused_rollout(name):
    """The same through rollout_kernel_v3 (tape-driven, unguarded trigonometry): two steps with a NaN / Inf action in the first one
    against the oracle stepping the same tape; the NaN state persists into the second step exactly as in the reference."""
    import torch
    from gym_amd.rollout import DeviceRollout

    n = 512
    rng = np.random.default_rng(5)
    r = DeviceRollout(GYM_IDS[name], n, seed=3, action_seed=4)
    ref = OracleEngine(name, n, LIMITS[name], seed=3, action_seed=4).o
    obs0 = r.reset(seed=3).cpu().numpy()
    robs0 = ref.reset(seed=3)
    assert ulps32(obs0, robs0).max() <= MAX_OBS_ULPS
    st, el = r.handle.get_state()
    ref.state[:] = st
    tape = rng.uniform(-1, 1, (3, n)).astype(np.float32)
    tape[0, ::3] = np.nan
    tape[0, 1::6] = np.inf
    tape[1, 4::6] = -np.inf
    out = r.rollout_tape(torch.from_numpy(tape).to(r.device))
    r.synchronize()
    with np.errstate(all="ignore"):
        for k in range(3):
            o, rw, te, tr, fin, fm = ref.step(tape[k])
            got_o, got_r = out["obs"][k].cpu().numpy(), out["reward"][k].cpu().numpy()
            assert np.array_equal(np.isnan(got_o), np.isnan(o)) and np.array_equal(np.isnan(got_r), np.isnan(rw)), f"{name} step {k}"
            ok = ~np.isnan(o)
            assert ulps32(got_o[ok], o[ok]).max() <= MAX_OBS_ULPS
            okr = ~np.isnan(rw)
            np.testing.assert_allclose(got_r[okr], rw[okr], rtol=REWARD_RTOL, atol=REWARD_ATOL.get(name, REWARD_ATOL_DEFAULT))
            assert np.array_equal(out["terminated"][k].cpu().numpy().astype(bool), te)
        assert np.isnan(o).any()
    r.close()


def test_huge_cartpole_velocities_inside_a_fused_rollout():
    """a FINITE injected theta_dot beyond ~1.3e154 overflows polemass_length * theta_dot^2 * sintheta; the unguarded fused
    kernel's reciprocal-based division would turn the infinite dividend into NaN where IEEE division — the reference — gives +-Inf.
    mxv_set_state sends such states (any CartPole velocity beyond 1e150) through the guarded launch: the final observations of the step
    hold the oracle's infinities, not NaNs, and the launch after it is the unguarded one again."""
    import torch
    from gym_amd.rollout import DeviceRollout

    n = 1024
    r = DeviceRollout("CartPole-v1", n, seed=3, action_seed=4)
    ref = OracleEngine("CartPole", n, 500, seed=3, action_seed=4).o
    r.reset(seed=3), ref.reset(seed=3)
    st, el = r.handle.get_state()
    st = st.copy()
    big = np.array([1.2e154, 1.4e154, 1e155, -1e160, 1e200, -1.7e308, 3e150, 1e151])
    st[3, : 8 * 16] = np.repeat(big, 16)                         # theta_dot: the overflowing product
    st[1, 200:264] = np.repeat(np.array([1e300, -1e300, 1e155, -2e200]), 16)      # x_dot: finite, beyond float32
    r.handle.set_state(st, el)
    ref.state[:] = st
    ref.elapsed[:] = el
    out = r.trajectory_buffers(2, layout="separate", want_final=True)
    r.rollout_per_step(2, out=out)
    r.synchronize()
    assert r.handle.last_launch()["safe"] == 1                    # the guarded instantiation took the injected state
    with np.errstate(all="ignore"):
        for k in range(2):
            a = ref.sample_actions()
            assert np.array_equal(a, out["actions"][k].cpu().numpy())
            o, rw, te, tr, fin, fm = ref.step(a)
            got_f, got_o = out["final_obs"][k].cpu().numpy(), out["obs"][k].cpu().numpy()
            assert np.array_equal(out["terminated"][k].cpu().numpy().astype(bool), te), k
            assert np.array_equal(np.isnan(got_f[fm]), np.isnan(fin[fm])) and np.array_equal(np.isinf(got_f[fm]), np.isinf(fin[fm])), k
            assert np.array_equal(got_f[fm][np.isinf(fin[fm])], fin[fm][np.isinf(fin[fm])])              # the sign of every infinity
            ok = np.isfinite(fin) & fm[:, None]
            assert (not ok.any() or ulps32(got_f[ok], fin[ok]).max() <= MAX_OBS_ULPS) and ulps32(got_o, o).max() <= MAX_OBS_ULPS
            if k == 0:
                assert np.isinf(fin[fm]).any() and te[:128].all()
    r.rollout_per_step(2, out=out)
    r.synchronize()
    assert r.handle.last_launch()["safe"] == 0
    r.close()


def test_nan_with_a_low_dword_payload_passes_the_clamps():
    """a NaN whose payload sits in the low dword only (0x7FF00000:00000001) is still a NaN after MountainCar's position and
    velocity clamps (np.clip propagates it) — splicing only the high dword onto a bound with a zero low dword made it an infinity."""
    from gym_amd import _native

    n = 64
    h = _native.Handle(ENV_IDS["MountainCar"], n, 200, seed=1, action_seed=2)
    h.reset_host()
    st, el = h.get_state()
    st = st.copy()
    weird = np.array([0x7FF0000000000001, 0xFFF0000000000001, 0x7FF00000FFFFFFFF], dtype=np.uint64).view(np.float64)
    assert np.isnan(weird).all()
    st[0, :3] = weird                       # position
    st[1, 8:11] = weird                     # velocity
    h.set_state(st, el)
    obs, rew, term, trunc, fin = h.step_host(np.ones(n, dtype=np.int64), want_final=True)
    after, _ = h.get_state()
    assert np.isnan(obs[:3]).all() and np.isnan(obs[8:11]).any(axis=1).all() and not np.isinf(obs).any()
    assert np.isnan(after[0, :3]).all() and np.isnan(after[1, 8:11]).all() and np.isfinite(after[:, 16:]).all()
    h.close()


def test_cartpole_beyond_marks_travel_with_a_snapshot():
    """the steps_beyond_terminated marks are part of a checkpoint — a restored NO_AUTORESET CartPole handle that had terminated
    keeps paying 0.0 (mxv_get_beyond / mxv_set_beyond; Handle.snapshot / restore), like the pickled reference env would."""
    from gym_amd import _native

    n = 256
    h = _native.Handle(ENV_IDS["CartPole"], n, 500, seed=5, action_seed=6, flags=_native.FLAG_NO_AUTORESET)
    h.reset_host()
    ones = np.ones(n, dtype=np.int64)
    for _ in range(40):                                            # always push right: every pole falls within ~10 steps
        obs, rew, term, trunc, _ = h.step_host(ones, want_final=False)
    assert term.all() and (rew == 0.0).all() and h.get_beyond().all()
    snap = h.snapshot()
    twin = _native.Handle(ENV_IDS["CartPole"], n, 500, seed=0, action_seed=0, flags=_native.FLAG_NO_AUTORESET)
    twin.reset_host()
    twin.restore(snap)
    assert twin.get_beyond().all()
    for _ in range(3):
        a, b = h.step_host(ones, want_final=False), twin.step_host(ones, want_final=False)
        assert (b[1] == 0.0).all() and all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a[:4], b[:4]))
    plain = _native.Handle(ENV_IDS["CartPole"], n, 500, seed=5, action_seed=6)
    assert not plain.get_beyond().any()
    plain.set_beyond(np.zeros(n, np.uint8))
    with pytest.raises(_native.MxvError):
        plain.set_beyond(np.ones(n, np.uint8))
    h.close(), twin.close(), plain.close()


def test_cartpole_stepped_on_after_termination():
    """MXV_FLAG_NO_AUTORESET keeps the reference's steps_beyond_terminated bookkeeping (cartpole.py:169-184): the fall pays 1.0, every
    later step that is still terminated 0.0, a reset clears the mark."""
    assert run_cartpole_beyond(HipEngine, strict=False) == 288


def test_known_answers_survey_appendix_b():
    from test_oracle_golden import kat_check

    kat_check(HipEngine, strict_bits=False)


def _rollout_compare(name, n, steps, seed, limit=None, env_offset=0, resync_every=1):
    """Device vs oracle on the full engine semantics: Philox-sampled actions, TimeLimit, autoreset from the Philox
    reset stream.  Both sides follow the same RNG contract, so actions and reset states are bit-identical and the
    two trajectories stay aligned; every `resync_every` steps the device's fp64 state is copied into the oracle so
    that last-bit libm differences cannot accumulate in the chaotic envs."""
    import torch
    from gym_amd import _native

    limit = LIMITS[name] if limit is None else limit
    h = _native.Handle(ENV_IDS[name], n, limit, seed=seed, action_seed=seed * 31 + 7, env_offset=env_offset)
    ref = OracleEngine(name, n, limit, seed=seed, action_seed=seed * 31 + 7, env_offset=env_offset).o
    obs0 = h.reset_host()
    robs0 = ref.reset(seed=seed)
    assert ulps32(obs0, robs0).max() <= MAX_OBS_ULPS
    dt = torch.int64 if DISCRETE[name] else torch.float32
    d_act = torch.zeros(n, dtype=dt, device="cuda")
    d_obs = torch.zeros((n, h.O), dtype=torch.float32, device="cuda")
    d_rew = torch.zeros(n, dtype=torch.float64, device="cuda")
    d_term = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_trunc = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_fin = torch.zeros((n, h.O), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ndone = 0
    for t in range(steps):
        st_pre, el_pre = h.get_state()
        if t % resync_every == 0:
            ref.state[:] = st_pre
            ref.elapsed[:] = el_pre
        want_a = ref.sample_actions()
        d_fin.zero_()
        torch.cuda.synchronize()
        h.step_sampled(d_obs, d_rew, d_term, d_trunc, d_fin, d_act)
        h.sync()
        robs, rrew, rterm, rtrunc, rfin, rfmask = ref.step(want_a)
        assert np.array_equal(d_act.cpu().numpy(), want_a), f"{name} t={t}: sampled actions differ"
        term, trunc = d_term.cpu().numpy().astype(bool), d_trunc.cpu().numpy().astype(bool)
        assert np.array_equal(term, rterm), f"{name} t={t}: terminated mask differs"
        assert np.array_equal(trunc, rtrunc), f"{name} t={t}: truncated mask differs"
        done = term | trunc
        obs, fin = d_obs.cpu().numpy(), d_fin.cpu().numpy()
        assert ulps32(obs, robs).max() <= MAX_OBS_ULPS, f"{name} t={t}: obs ulps {ulps32(obs, robs).max()}"
        np.testing.assert_allclose(obs, robs, rtol=OBS_RTOL, atol=1e-30)
        if done.any():
            assert ulps32(fin[done], rfin[done]).max() <= MAX_OBS_ULPS
            assert np.all(fin[~done] == 0), "final_obs rows of unfinished envs must stay untouched"
        np.testing.assert_allclose(d_rew.cpu().numpy(), rrew, rtol=REWARD_RTOL, atol=REWARD_ATOL.get(name, REWARD_ATOL_DEFAULT))
        st, el = h.get_state()
        assert np.array_equal(el, ref.elapsed)
        assert np.array_equal(st[:, done], ref.state[:, done]), "post-reset states are pure Philox: must be bit-exact"
        # free-running stretches (resync_every > 1): last-bit differences grow with the dynamics' own sensitivity
        np.testing.assert_allclose(st, ref.state, rtol=1e-12 if resync_every == 1 else 1e-9, atol=1e-13 if resync_every == 1 else 1e-10)
        ndone += int(done.sum())
    return ndone


@pytest.mark.parametrize("name", ENV_NAMES)
def test_rollout_vs_oracle_default_limits(name):
    ndone = _rollout_compare(name, n=3000, steps=260, seed=1234, env_offset=4096)
    if name in ("CartPole", "Pendulum", "MountainCar"):
        assert ndone > 0


@pytest.mark.parametrize("name", ENV_NAMES)
def test_rollout_vs_oracle_short_limit(name):
    ndone = _rollout_compare(name, n=1025, steps=60, seed=99, limit=7)
    assert ndone >= 1025 * 8


@pytest.mark.parametrize("name,steps,resync", [("CartPole", 300, 10 ** 9), ("MountainCar", 260, 10 ** 9), ("MountainCarContinuous", 200, 16),
                                               ("Pendulum", 220, 16), ("Acrobot", 64, 8)])
def test_rollout_vs_oracle_multi_step_drift(name, steps, resync):
    """The per-step comparisons above hand the device's fp64 state to the oracle before EVERY step; here the two run free for
    `resync` steps at a time (CartPole and MountainCar: never resynchronised — only their own autoresets, whose Philox states
    are bit-identical on both sides, realign them), so last-bit libm differences must stay below the bars over whole episodes:
    masks and sampled actions exact, observations <= 2 float32 ulps, fp64 state rtol 1e-12."""
    ndone = _rollout_compare(name, n=4096, steps=steps, seed=77, env_offset=8192, resync_every=resync)
    if name in ("CartPole", "MountainCar", "Pendulum"):
        assert ndone >= 4096


def test_config1_cartpole_8_envs_1000_steps():
    """BASELINE.json configs[0]: CartPole-v1, num_envs=8, random actions, 1000 steps (plumbing case)."""
    ndone = _rollout_compare("CartPole", n=8, steps=1000, seed=0)
    assert ndone > 100  # random-policy episodes last ~22 steps


@pytest.mark.parametrize("name,n", [("CartPole", 1 << 20), ("Pendulum", 1 << 19), ("MountainCarContinuous", 1 << 19),
                                    ("Acrobot", 1 << 19), ("MountainCar", 1 << 18)])
def test_full_size_against_oracle_and_properties(name, n):
    """BASELINE.json sizes: 16 steps of the whole batch against the oracle, then invariants over a longer run."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from helpers import GYM_IDS

    steps = 12 if name == "Acrobot" else 16
    _rollout_compare(name, n=n, steps=steps, seed=5, resync_every=1)

    def run(env_offset, count, graph):
        r = DeviceRollout(GYM_IDS[name], count, env_offset=env_offset, seed=77, action_seed=78)
        r.reset(seed=77)
        with torch.cuda.stream(r.stream):
            acc = torch.zeros(count, dtype=torch.float64, device="cuda")
            dones = torch.zeros(count, dtype=torch.int64, device="cuda")
        for _ in range(3):
            r.rollout(20, mode=graph)
            with torch.cuda.stream(r.stream):
                acc += r.reward
                dones += (r.terminated | r.truncated).to(torch.int64)
        r.synchronize()
        st, el = r.handle.get_state()
        out = (r.obs.cpu().numpy().copy(), acc.cpu().numpy(), dones.cpu().numpy(), st, el)
        r.close()
        return out

    whole = run(0, n, "eager")
    for other in ("graph", "fused"):  # determinism; hipGraph replay == eager launches == fused K-step launch
        again = run(0, n, other)
        for a, b in zip(whole, again):
            assert np.array_equal(a, b), other
    lo, hi = run(0, n // 2, "fused"), run(n // 2, n // 2, "fused")  # shard-invariance (multi-GPU partition rule)
    assert np.array_equal(whole[0], np.concatenate([lo[0], hi[0]]))
    assert np.array_equal(whole[3], np.concatenate([lo[3], hi[3]], axis=1))
    assert np.array_equal(whole[4], np.concatenate([lo[4], hi[4]]))
    obs, _, _, st, el = whole
    assert np.all(np.isfinite(obs)) and np.all(np.isfinite(st))
    assert el.min() >= 0 and el.max() < LIMITS[name]
    from gym_amd.registration import single_spaces, spec
    space, _ = single_spaces(spec(GYM_IDS[name]).kind)
    assert np.all(obs >= space.low - 1e-6) and np.all(obs <= space.high + 1e-6)


def test_invalid_action_is_latched_and_reported():
    from gym_amd import _native

    h = _native.Handle(ENV_IDS["CartPole"], 64, 500, seed=1)
    h.reset_host()
    st0, el0 = h.get_state()
    bad = np.zeros(64, dtype=np.int64)
    bad[17] = 2
    with pytest.raises(_native.MxvError) as ei:
        h.step_host(bad)
    assert ei.value.code == _native.ERR_INVALID_ACTION
    st1, el1 = h.get_state()
    assert np.array_equal(st0[:, 17], st1[:, 17]) and el1[17] == el0[17]  # the offending env was not stepped
    h.step_host(np.zeros(64, dtype=np.int64))  # latch cleared, engine usable again


def test_step_before_reset_is_refused():
    from gym_amd import _native

    h = _native.Handle(ENV_IDS["Pendulum"], 16, 200)
    with pytest.raises(_native.MxvError) as ei:
        h.step_host(np.zeros(16, dtype=np.float32))
    assert ei.value.code == _native.ERR_RESET_NEEDED


def test_non_default_params_match_oracle():
    """set_attr path: the runtime-parameter kernels against the oracle with the same parameter vector."""
    from gym_amd import _native

    rng = np.random.default_rng(0)
    tweaks = {"CartPole": {0: 11.0, 6: 12.5, 10: 1.0}, "Pendulum": {3: 9.81, 1: 1.5}, "Acrobot": {11: 1.0, 3: 1.2},
              "MountainCar": {4: 0.01, 5: 0.0012}, "MountainCarContinuous": {7: 0.002, 6: 0.005}}
    for name in ENV_NAMES:
        g = np.load(f"{__import__('helpers').GOLDEN}/{name}_p1.npz")
        n = 2048
        eng = HipEngine(name, n, 0, autoreset=False)
        orc = OracleEngine(name, n, 0, autoreset=False)
        p = eng.h.get_params()
        for k, v in tweaks[name].items():
            p[k] = v
        eng.h.set_params(p)
        orc.o.P[:] = p
        el = np.where(g["fresh"][:n] == 1, 0, 3).astype(np.int32)
        eng.set_state(g["state0"][:n].T, el)
        orc.set_state(g["state0"][:n].T, el)
        a = g["action"][:n]
        o1, r1, t1, _, _ = eng.step(a)
        o2, r2, t2, _, _ = orc.step(a)
        assert np.array_equal(t1, t2), name
        assert ulps32(o1, o2).max() <= MAX_OBS_ULPS, name
        np.testing.assert_allclose(r1, r2, rtol=REWARD_RTOL, atol=REWARD_ATOL.get(name, REWARD_ATOL_DEFAULT))
        np.testing.assert_allclose(eng.get_state()[0], orc.get_state()[0], rtol=1e-12, atol=1e-13)


def test_reward_f32_and_action_i32_flags():
    import torch
    from gym_amd import _native

    n = 4096
    h = _native.Handle(ENV_IDS["Acrobot"], n, 500, seed=3, action_seed=4,
                       flags=_native.FLAG_REWARD_F32 | _native.FLAG_ACTION_I32)
    h2 = _native.Handle(ENV_IDS["Acrobot"], n, 500, seed=3, action_seed=4)
    h.reset_host(), h2.reset_host()
    a64 = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    h2.sample_actions(a64)
    h2.sync()
    a = a64.cpu().numpy()
    o1, r1, t1, tr1, _ = h.step_host(a.astype(np.int32))
    o2, r2, t2, tr2, _ = h2.step_host(a)
    assert r1.dtype == np.float32 and np.array_equal(o1, o2) and np.array_equal(r1, r2.astype(np.float32))


@pytest.mark.parametrize("name", ENV_NAMES)
def test_fused_rollout_equals_single_steps_and_oracle(name):
    """mxv_rollout(FUSED) and mxv_rollout_tape (one launch, K steps, state in registers) against K single-step
    launches (bit-exact: same kernel body) and against the oracle stepping the recorded action tape."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from helpers import GYM_IDS

    n, K = 2500, 37
    a = DeviceRollout(GYM_IDS[name], n, seed=21, action_seed=22, env_offset=512)
    b = DeviceRollout(GYM_IDS[name], n, seed=21, action_seed=22, env_offset=512)
    c = DeviceRollout(GYM_IDS[name], n, seed=21, action_seed=22, env_offset=512)
    for r in (a, b, c):
        r.reset(seed=21)
    st0, el0 = a.handle.get_state()
    fused = a.rollout_per_step(K, mode="fused")
    eager = b.rollout_per_step(K, mode="eager")
    a.synchronize(), b.synchronize()
    for key in ("obs", "reward", "terminated", "truncated", "actions"):
        assert torch.equal(fused[key], eager[key]), key
    taped = c.rollout_tape(fused["actions"])
    c.synchronize()
    for key in ("obs", "reward", "terminated", "truncated"):
        assert torch.equal(fused[key], taped[key]), key
    for x, y in zip(a.handle.get_state(), b.handle.get_state()):
        assert np.array_equal(x, y)
    assert a.handle.get_counters() == b.handle.get_counters() == (K, 1)
    # oracle on the same tape; resynchronised to the device's trajectory only through the recorded outputs
    ref = OracleEngine(name, n, LIMITS[name], seed=21, action_seed=22, env_offset=512).o
    ref.reset(seed=21)
    assert np.array_equal(ref.state, st0)
    acts = fused["actions"].cpu().numpy()
    obs, term, trunc = fused["obs"].cpu().numpy(), fused["terminated"].cpu().numpy(), fused["truncated"].cpu().numpy()
    horizon = K if name in ("CartPole", "MountainCar", "MountainCarContinuous") else 12  # chaotic envs: short horizon
    for k in range(horizon):
        assert np.array_equal(ref.sample_actions(), acts[k])
        robs, rrew, rterm, rtrunc, _, _ = ref.step(acts[k])
        assert np.array_equal(term[k].astype(bool), rterm) and np.array_equal(trunc[k].astype(bool), rtrunc), (name, k)
        np.testing.assert_allclose(obs[k], robs, rtol=OBS_RTOL, atol=1e-30)
    for r in (a, b, c):
        r.close()


@pytest.mark.parametrize("limit", [1, 2, 5])
@pytest.mark.parametrize("name", ENV_NAMES)
def test_fused_rollout_with_tiny_time_limits_equals_single_steps(name, limit):
    """rollout_kernel_v3 keeps every env's NEXT reset ready in a lane-private LDS slot, refilled by look-ahead passes every few
    steps; an env that finishes again before its pass came round (TimeLimit of 1, 2, 5 steps) must force the refill early.
    Several launches back to back (slots are rebuilt at every kernel entry), starting at a step index that is not a multiple
    of 32 (Discrete(2) bit blocks) and crossing several blocks / ring refills; n is not a multiple of the tile."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from helpers import GYM_IDS

    n = 2111
    a = DeviceRollout(GYM_IDS[name], n, seed=31, action_seed=32, env_offset=256, max_episode_steps=limit)
    b = DeviceRollout(GYM_IDS[name], n, seed=31, action_seed=32, env_offset=256, max_episode_steps=limit)
    a.reset(seed=31), b.reset(seed=31)
    for K in (7, 70, 33):
        fa = a.rollout_per_step(K, mode="fused")
        fb = b.rollout_per_step(K, mode="eager")
        a.synchronize(), b.synchronize()
        for key in ("obs", "reward", "terminated", "truncated", "actions"):
            assert torch.equal(fa[key], fb[key]), (K, key)
        assert fa["truncated"].sum().item() >= n * (K // limit - 1)
    for x, y in zip(a.handle.get_state(), b.handle.get_state()):
        assert np.array_equal(x, y)
    ea, eb = a.handle.get_episodes(), b.handle.get_episodes()
    assert np.array_equal(ea, eb) and ea.min() >= 1 + 110 // limit - 1
    a.close(), b.close()


def test_reset_ordinals_follow_each_env_and_survive_a_checkpoint():
    """RNG contract: env i's k-th reset draws Philox(seed_i, (k, 0, 0, 2<<28)).  The ordinals are device state: they advance
    with explicit resets (masked ones only for the masked envs) and autoresets, restart at mxv_seed, and travel with
    snapshot()/restore()."""
    from gym_amd import _native

    n = 512
    h = _native.Handle(ENV_IDS["CartPole"], n, 4, seed=5, action_seed=6)
    h.reset_host()
    assert np.all(h.get_episodes() == 1)
    mask = (np.arange(n) % 4 == 0).astype(np.uint8)
    h.reset_host(mask=mask)
    assert np.array_equal(h.get_episodes(), 1 + mask)
    for _ in range(8):                       # TimeLimit 4: everything autoresets at steps 4 and 8 (and some terminate earlier)
        h.step_host(np.zeros(n, np.int64))
    ep = h.get_episodes()
    assert np.all(ep >= 3 + mask)
    snap = h.snapshot()
    g = _native.Handle(ENV_IDS["CartPole"], n, 4, seed=99, action_seed=98)
    g.restore(snap)
    assert np.array_equal(g.get_episodes(), ep)
    for _ in range(9):
        o1 = h.step_host(np.ones(n, np.int64))
        o2 = g.step_host(np.ones(n, np.int64))
        assert all(np.array_equal(x, y) for x, y in zip(o1[:4], o2[:4]))
        done = o1[2] | o1[3]
        assert np.array_equal(o1[4][done], o2[4][done])      # final_obs rows exist only for the envs that finished
    h.seed(5)
    assert np.all(h.get_episodes() == 0)
    h.close(), g.close()


def test_acrobot_torque_noise_matches_oracle_twin():
    """acrobot.py:202-205 `torque += np_random.uniform(-torque_noise_max, torque_noise_max)`: the noise comes from the
    engine's Philox step-noise stream (one word per env-step, keyed by the env's seed), so device and oracle twin stay
    aligned step by step; the arithmetic itself is pinned against the reference by tests/golden/Acrobot_noise_p1.npz
    (oracle, bit-exact).  Broadcast value, then per-env values (noise on the even envs only); a K-step launch (routed to
    the per-step kernel) equals K single steps; noise really changes the trajectories."""
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    n, steps = 2048, 24
    for per_env in (False, True):
        h = _native.Handle(ENV_IDS["Acrobot"], n, 40, seed=5, action_seed=6)
        noisy = OracleEngine("Acrobot", n, 40, seed=5, action_seed=6).o
        quiet = OracleEngine("Acrobot", n, 40, seed=5, action_seed=6).o
        p = h.get_params()
        nm = 0.7 if per_env else 0.35
        noisy.P[10] = nm
        if per_env:
            table = np.repeat(p[:, None], n, axis=1)
            table[10, ::2] = nm
            h.set_params_per_env(table)
        else:
            p[10] = nm
            h.set_params(p)
        pick = (np.arange(n) % 2 == 0) if per_env else np.ones(n, bool)   # envs that follow the noisy oracle
        h.reset_host()
        for o in (noisy, quiet):
            o.reset(seed=5)
        rng = np.random.default_rng(1)
        for t in range(steps):
            st, el = h.get_state()
            for o in (noisy, quiet):
                o.state[:], o.elapsed[:] = st, el
            a = rng.integers(0, 3, n)
            obs, rew, term, trunc, fin = h.step_host(a)
            rn, rq = noisy.step(a), quiet.step(a)
            robs = np.where(pick[:, None], rn[0], rq[0])
            assert np.array_equal(term, np.where(pick, rn[2], rq[2])) and np.array_equal(trunc, np.where(pick, rn[3], rq[3]))
            assert ulps32(obs, robs).max() <= MAX_OBS_ULPS
            np.testing.assert_allclose(h.get_state()[0], np.where(pick[None, :], noisy.state, quiet.state), rtol=1e-12, atol=1e-13)
            assert not np.array_equal(rn[0], rq[0])                       # the noise is not a no-op
        h.close()
    outs = []
    for K in (1, 12):
        r = DeviceRollout("Acrobot-v1", n, seed=5, action_seed=6)
        p = r.handle.get_params()
        p[10] = 0.35
        r.handle.set_params(p)
        r.reset(seed=5)
        tr = r.trajectory_buffers(12)
        if K == 1:
            for k in range(12):
                r.handle.rollout(1, tr["obs"][k], tr["reward"][k], tr["terminated"][k], tr["truncated"][k], None, tr["actions"][k])
        else:
            r.rollout_per_step(12, out=tr)
        r.synchronize()
        outs.append(tr["obs"].cpu().numpy().copy())
        r.close()
    assert np.array_equal(outs[0], outs[1])
    q = DeviceRollout("Acrobot-v1", n, seed=5, action_seed=6)
    q.reset(seed=5)
    qo = q.rollout_per_step(12)["obs"].cpu().numpy()
    q.close()
    assert not np.array_equal(qo[0], outs[0][0])


@pytest.mark.parametrize("name", ["CartPole", "Pendulum"])
def test_trajectory_final_observations(name):
    """rollout_per_step(want_final): final_obs[k, i] holds info["final_observation"] of step k for the envs that finished
    there (sync_vector_env.py:152-156) and is untouched elsewhere; fused == per-step launches; the terminal observation
    differs from the returned (post-reset) one."""
    from gym_amd.rollout import DeviceRollout

    n, K, limit = 3000, 40, 9
    res = {}
    for mode in ("fused", "eager"):
        r = DeviceRollout(GYM_IDS_LOCAL[name], n, seed=3, action_seed=4, max_episode_steps=limit)
        r.reset(seed=3)
        tr = r.trajectory_buffers(K, want_final=True)
        r.rollout_per_step(K, mode=mode, out=tr)
        r.synchronize()
        res[mode] = {k: v.cpu().numpy() for k, v in tr.items()}
        r.close()
    a, b = res["fused"], res["eager"]
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    done = (a["terminated"] | a["truncated"]).astype(bool)
    assert done.any() and np.all(a["final_obs"][~done] == 0)
    assert np.all(np.any(a["final_obs"][done] != a["obs"][done], axis=1))
    if name == "CartPole":   # a terminated CartPole's terminal observation is outside the thresholds, the reset one inside
        term = a["terminated"].astype(bool)
        fo = a["final_obs"][term]
        assert np.all((np.abs(fo[:, 0]) > 2.4) | (np.abs(fo[:, 2]) > 12 * 2 * np.pi / 360))
        assert np.all(np.abs(a["obs"][term]) <= 0.05)


GYM_IDS_LOCAL = {"CartPole": "CartPole-v1", "Pendulum": "Pendulum-v1"}


@pytest.mark.gpu
@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("name", ENV_NAMES)
def test_tape_driven_rollout_equals_stepping_with_the_same_actions(name, compact):
    """mxv_rollout_tape runs rollout_kernel_v3 with the action words replaced by a caller's [K][N] tape, read two steps ahead
    (loop unrolled by two, a second loop instantiation for waves whose env slots are all real).  Against one step_kernel launch
    per step fed the same rows: every output of every step, final observations, then state / elapsed steps / reset ordinals,
    bit for bit — over launches with even, odd and one-step-pair K, a tiny TimeLimit (ready-made reset entries run out), n not a
    multiple of the tile (the last wave takes the other loop), and both dtype sets."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from helpers import GYM_IDS

    n, limit = 5000 + 77, 6
    kw = dict(seed=41, action_seed=42, env_offset=1 << 21, max_episode_steps=limit, reward_f32=compact, action_i32=compact)
    src = DeviceRollout(GYM_IDS[name], n, **kw)     # only produces action rows (its own Philox stream)
    a = DeviceRollout(GYM_IDS[name], n, **kw)
    b = DeviceRollout(GYM_IDS[name], n, **kw)
    for r in (src, a, b):
        r.reset(seed=41)
    ndone = 0
    for K in (2, 9, 64, 33):
        rows = src.rollout_per_step(K, mode="fused")["actions"]
        src.synchronize()                      # the rows are written on src's stream; clone() runs on torch's current one
        tape = rows.clone()
        out = a.rollout_tape(tape, out=a.trajectory_buffers(K, want_final=True))
        a.synchronize()
        for k in range(K):
            o, r, te, tr = b.step(tape[k], want_final=True)
            b.synchronize()
            assert torch.equal(out["obs"][k], o) and torch.equal(out["reward"][k], r), (name, K, k)
            assert torch.equal(out["terminated"][k], te) and torch.equal(out["truncated"][k], tr), (name, K, k)
            done = (te | tr).bool()
            assert torch.equal(out["final_obs"][k][done], b.final_obs[done]), (name, K, k)
            ndone += int(done.sum())
        for x, y in zip(a.handle.get_state(), b.handle.get_state()):
            assert np.array_equal(x, y)
        assert np.array_equal(a.handle.get_episodes(), b.handle.get_episodes())
        assert a.handle.get_counters() == b.handle.get_counters()
    assert ndone >= 10 * n
    # the trajectory shape without final observations (the straight-line instantiation) gives the same numbers
    rows = src.rollout_per_step(16, mode="fused")["actions"]
    src.synchronize()
    tape = rows.clone()
    st = a.handle.snapshot()
    full = a.rollout_tape(tape, out=a.trajectory_buffers(16, want_final=True))
    a.synchronize()
    b.handle.restore(st)
    plain = b.rollout_tape(tape)
    b.synchronize()
    for key in ("obs", "reward", "terminated", "truncated"):
        assert torch.equal(full[key], plain[key]), key
    for r in (src, a, b):
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["CartPole", "Acrobot", "MountainCar"])
def test_tape_with_an_out_of_range_action_latches_the_error_and_leaves_that_env_alone(name):
    """Discrete.contains (cartpole.py:131-132) inside a tape: the launch latches the error (raised at the next synchronisation,
    as for a single step), the offending env's state is not written back, every other env is stepped as if nothing happened."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from helpers import GYM_IDS

    n, K, victim = 1000, 12, 333
    a = DeviceRollout(GYM_IDS[name], n, seed=51, action_seed=52)
    b = DeviceRollout(GYM_IDS[name], n, seed=51, action_seed=52)
    a.reset(seed=51), b.reset(seed=51)
    rows = b.rollout_per_step(K, mode="fused")["actions"]
    b.synchronize()
    tape = rows.clone()
    st0, el0 = a.handle.get_state()
    bad = tape.clone()
    bad[5, victim] = 7
    out = a.rollout_tape(bad)
    with pytest.raises(AssertionError):
        a.synchronize()
    st1, el1 = a.handle.get_state()
    stb, elb = b.handle.get_state()
    keep = np.arange(n) != victim
    assert np.array_equal(st1[:, keep], stb[:, keep]) and np.array_equal(el1[keep], elb[keep])
    assert np.array_equal(st1[:, victim], st0[:, victim]) and el1[victim] == el0[victim]
    a.close(), b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("autoreset", [True, False])
@pytest.mark.parametrize("params", ["default", "broadcast", "per_env"])
@pytest.mark.parametrize("name", ["CartPole", "Pendulum", "Acrobot"])
def test_tape_dispatch_matrix_equals_stepping(name, params, autoreset):
    """mxv_rollout_tape takes the fused rollout kernel only for default physics parameters with autoreset; changed parameters
    (one value for all envs, or one per env) and MXV_FLAG_NO_AUTORESET run step_kernel's K-loop.  Whichever kernel serves it, a
    tape gives what K single steps with the same rows give, bit for bit."""
    import torch
    from gym_amd.rollout import DeviceRollout
    from helpers import GYM_IDS

    n, K = 3000, 21
    kw = dict(seed=71, action_seed=72, max_episode_steps=8, autoreset=autoreset)
    src = DeviceRollout(GYM_IDS[name], n, seed=71, action_seed=72)
    a, b = DeviceRollout(GYM_IDS[name], n, **kw), DeviceRollout(GYM_IDS[name], n, **kw)
    for r in (a, b):
        if params == "broadcast":
            p = r.handle.get_params()
            p[0] *= 1.07                    # gravity (CartPole, Acrobot's g analogue) / max_speed ... any attribute: only != default matters
            r.handle.set_params(p)
        elif params == "per_env":
            t = r.handle.get_params_per_env()
            t[0] *= np.linspace(0.9, 1.1, n)
            r.handle.set_params_per_env(t)
    src.reset(seed=71), a.reset(seed=71), b.reset(seed=71)
    rows = src.rollout_per_step(K, mode="fused")["actions"]
    src.synchronize()
    tape = rows.clone()
    out = a.rollout_tape(tape, out=a.trajectory_buffers(K, want_final=True))
    a.synchronize()
    for k in range(K):
        o, r, te, tr = b.step(tape[k], want_final=True)
        b.synchronize()
        assert torch.equal(out["obs"][k], o) and torch.equal(out["reward"][k], r), (k,)
        assert torch.equal(out["terminated"][k], te) and torch.equal(out["truncated"][k], tr), (k,)
    for x, y in zip(a.handle.get_state(), b.handle.get_state()):
        assert np.array_equal(x, y)
    for r in (src, a, b):
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ENV_NAMES)
def test_non_default_physics_attributes_against_the_reference(name):
    """<env>_p1_variants.npz (make_golden_variants.py: the live reference stepped with changed attributes — semi-implicit
    CartPole, nips Acrobot, Pendulum g, goal_velocity, moved thresholds, masses, lengths, time steps): the device's
    runtime-parameter kernels against the reference's own outputs, masks exact, observations within the usual bars."""
    from helpers import run_p1_variants

    run_p1_variants(HipEngine, name, strict=False)
