"""SURVEY.md §8(f)-2, the per-sub-env form on the device: mxv_subnorm_* (gym_amd/csrc/mxv_subnorm.hip) — what
`gym.vector.make(id, n, wrappers=[NormalizeObservation, NormalizeReward])` keeps for every sub-env (gym/vector/__init__.py:56-65 around
gym/wrappers/normalize.py:50-145) — against the oracle (oracle/normalize.c: orc_subnorm_*, pinned bit for bit to the reference's own run by
tests/test_normalize_oracle.py) and against the reference's golden through gym_amd.make(wrappers=...)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    """Equal bit patterns, NaN matching NaN (x86 and gfx950 sign their default NaNs differently)."""
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and np.array_equal(a[~nan].view(np.uint64), b[~nan].view(np.uint64))


def _case(rng, K, n, D, p_done=0.08):
    x = (rng.standard_normal((K, n, D)) * rng.uniform(0.01, 6.0, (1, 1, D))).astype(np.float32)
    fin = (rng.standard_normal((K, n, D)) * 3).astype(np.float32)
    te = (rng.random((K, n)) < p_done).astype(np.uint8)
    tr = (rng.random((K, n)) < p_done / 2).astype(np.uint8)
    return x, fin, te, tr


@pytest.mark.parametrize("D", [1, 2, 3, 4, 6])
@pytest.mark.parametrize("n", [1, 1000, 4099])
def test_observation_kernel_equals_the_oracle_bit_for_bit(D, n):
    """Every supported dim, ragged sizes, a reset() call, then single steps and a 7-step trajectory call, float32 and float64 results,
    terminal rows where an episode ended: statistics, batched rows and float64 final rows equal the oracle's, bit for bit."""
    import torch

    from gym_amd import _native
    from oracle.oracle import SubEnvNorm

    rng = np.random.default_rng(100 * D + n)
    dev = torch.device("cuda", 0)
    d, o = _native.SubNorm(D, n), SubEnvNorm(D, n)
    x0 = rng.standard_normal((1, n, D)).astype(np.float32)
    y_d = torch.zeros((1, n, D), dtype=torch.float32, device=dev)
    d.observations(1, torch.from_numpy(x0).to(dev), None, None, None, y_d, True, None, 1e-8)
    y_o = np.zeros((1, n, D), np.float32)
    o.observations(1, x0, None, None, None, y_o, True, None, 1e-8)
    assert np.array_equal(y_d.cpu().numpy(), y_o)
    for K, f32 in ((1, True), (7, True), (1, False), (5, False)):
        x, fin, te, tr = _case(rng, K, n, D)
        dt = torch.float32 if f32 else torch.float64
        y_d = torch.zeros((K, n, D), dtype=dt, device=dev)
        yf_d = torch.zeros((K, n, D), dtype=torch.float64, device=dev)
        d.observations(K, torch.from_numpy(x).to(dev), torch.from_numpy(fin).to(dev), torch.from_numpy(te).to(dev), torch.from_numpy(tr).to(dev),
                       y_d, f32, yf_d, 1e-8)
        y_o, yf_o = np.zeros((K, n, D), np.float32 if f32 else np.float64), np.zeros((K, n, D))
        o.observations(K, x, fin, te, tr, y_o, f32, yf_o, 1e-8)
        done = (te | tr).astype(bool)
        assert np.array_equal(y_d.cpu().numpy(), y_o), (K, f32)
        assert np.array_equal(yf_d.cpu().numpy()[done], yf_o[done]), (K, f32)
        assert done.any() or n == 1
    for a, b in zip(d.get_state()[:3], o.get_state()[:3]):
        assert np.array_equal(a, b)
    d.close()


def test_extreme_statistics_take_the_plain_division_and_stay_bit_exact():
    """The kernels share one reciprocal per update where that reproduces IEEE division (ordinary operands) and fall back to `/` per wave
    otherwise: injected statistics with tiny / huge / zero / non-finite entries (dividends below 2^-723, counts of 1e-300 and 1e300, NaN)
    sit in the same waves as ordinary sub-envs and every sub-env still equals the oracle bit for bit."""
    import torch

    from gym_amd import _native
    from oracle.oracle import SubEnvNorm

    n, D = 777, 4
    rng = np.random.default_rng(9)
    dev = torch.device("cuda", 0)
    mean = rng.standard_normal((n, D))
    var = rng.random((n, D)) + 0.05
    count = rng.random(n) * 100 + 1
    weird = rng.permutation(n)[:60]
    mean[weird[:10]] = 1e300
    mean[weird[10:20], 1] = 1e-300
    var[weird[20:30]] = 1e-310
    var[weird[30:35], 2] = np.inf
    count[weird[35:45]] = 1e-300
    count[weird[45:50]] = 1e300
    mean[weird[50:55], 0] = np.nan
    count[weird[55:60]] = -1.0            # tot = 0
    d, o = _native.SubNorm(D, n), SubEnvNorm(D, n)
    d.set_state(mean, var, count), o.set_state(mean, var, count)
    with np.errstate(all="ignore"):
        for K in (1, 6):
            x, fin, te, tr = _case(rng, K, n, D)
            x[:, weird[:3]] = 0.0
            y_d = torch.zeros((K, n, D), dtype=torch.float64, device=dev)
            yf_d = torch.zeros((K, n, D), dtype=torch.float64, device=dev)
            d.observations(K, torch.from_numpy(x).to(dev), torch.from_numpy(fin).to(dev), torch.from_numpy(te).to(dev), torch.from_numpy(tr).to(dev),
                           y_d, False, yf_d, 1e-8)
            y_o, yf_o = np.zeros((K, n, D)), np.zeros((K, n, D))
            o.observations(K, x, fin, te, tr, y_o, False, yf_o, 1e-8)
            assert _same_bits(y_d.cpu().numpy(), y_o), K
    for a, b in zip(d.get_state()[:3], o.get_state()[:3]):
        assert _same_bits(a, b)
    d.close()


@pytest.mark.parametrize("n", [3, 5000])
def test_reward_kernel_equals_the_oracle_bit_for_bit(n):
    import torch

    from gym_amd import _native
    from oracle.oracle import SubEnvNorm

    rng = np.random.default_rng(n)
    dev = torch.device("cuda", 0)
    for f32 in (False, True):
        d, o = _native.SubNorm(1, n), SubEnvNorm(1, n)
        for K in (1, 9, 1):
            rew = rng.standard_normal((K, n)) * 3
            if f32:
                rew = rew.astype(np.float32)
            te = (rng.random((K, n)) < 0.1).astype(np.uint8)
            tr = (rng.random((K, n)) < 0.05).astype(np.uint8)
            r_d = torch.from_numpy(rew).to(dev)
            d.rewards(K, r_d, f32, torch.from_numpy(te).to(dev), torch.from_numpy(tr).to(dev), r_d, 0.97, 1e-8)     # in place
            out = np.zeros((K, n), rew.dtype)
            o.rewards(K, rew, f32, te, tr, out, 0.97, 1e-8)
            assert np.array_equal(r_d.cpu().numpy(), out), (f32, K)
        for a, b in zip(d.get_state(), o.get_state()):
            assert np.array_equal(a, b)
        d.close()


def test_state_round_trip_and_errors():
    import torch

    from gym_amd import _native

    d = _native.SubNorm(4, 10)
    mean, var, count, ret = d.get_state()
    assert (mean == 0).all() and (var == 1).all() and (count == 1e-4).all() and (ret == 0).all()      # RunningMeanStd.__init__, normalize.py:12-15
    rng = np.random.default_rng(0)
    m, v, c, r = rng.standard_normal((10, 4)), rng.random((10, 4)) + 0.1, rng.random(10) * 50 + 1, rng.standard_normal(10)
    d.set_state(m, v, c, r)
    for a, b in zip(d.get_state(), (m, v, c, r)):
        assert np.array_equal(a, b)
    x = torch.zeros((1, 10, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(_native.MxvError):
        d.observations(0, x, None, None, None, x, True, None, 1e-8)                      # K <= 0
    with pytest.raises(_native.MxvError):
        d.observations(1, x, None, None, None, x, False, None, 1e-8)                     # float64 results may not alias the float32 input
    with pytest.raises(_native.MxvError):
        d.rewards(1, x, True, x, x, x, 0.99, 1e-8)                                       # dim != 1
    with pytest.raises(_native.MxvError):
        _native.SubNorm(5, 10)                                                           # unsupported dim
    d.close()


@pytest.mark.parametrize("name", ["CartPole", "Pendulum"])
def test_vector_make_normalize_wrappers_on_the_device_against_the_reference(name):
    """The reference's own run of make(wrappers=[TimeLimit, NormalizeObservation, NormalizeReward, RecordEpisodeStatistics]) (golden)
    replayed with every sub-env's statistics ON THE DEVICE (device=True: what make() picks from SUBENV_DEVICE_MIN sub-envs on), held to
    helpers.subnorm_bounds — the engine's raw-output bars propagated through the statistics (derived, not picked)."""
    from helpers import replay_vector_make_normalize

    assert replay_vector_make_normalize(name, exact=False, device=True) > 50


def test_make_picks_the_device_from_the_threshold_on_and_both_paths_agree_bit_for_bit():
    """gym_amd.make(..., wrappers=[NormalizeObservation, NormalizeReward]) with >= SUBENV_DEVICE_MIN sub-envs keeps the statistics on the
    device and reads the step's outputs where the host step left them (staging); forced to the host (device=False) over an identically
    seeded engine it returns the same bits: batched observations, rewards, float64 final observations, statistics."""
    import gym_amd
    from gym_amd import wrappers as W

    n = W.SUBENV_DEVICE_MIN
    a = gym_amd.make("CartPole-v1", num_envs=n, wrappers=[W.NormalizeObservation, W.NormalizeReward])
    assert type(a).__name__ == "SubEnvNormalizeReward" and a._sub is not None and a.env._sub is not None and a.env._staged
    b = W.SubEnvNormalizeReward(W.SubEnvNormalizeObservation(gym_amd.make("CartPole-v1", num_envs=n), device=False), device=False)
    oa, _ = a.reset(seed=5)
    ob, _ = b.reset(seed=5)
    assert oa.dtype == np.float32 and np.array_equal(oa, ob)
    a.action_space.seed(1)
    finished = 0
    for t in range(40):
        act = a.action_space.sample()
        xa, xb = a.step(act), b.step(act)
        for u, v in zip(xa[:4], xb[:4]):
            assert u.dtype == v.dtype and np.array_equal(u, v), t
        done = xa[2] | xa[3]
        for i in np.flatnonzero(done)[:50]:
            fa, fb = xa[4]["final_observation"][i], xb[4]["final_observation"][i]
            assert fa.dtype == np.float64 and np.array_equal(fa, fb)
        finished += int(done.sum())
    assert finished > 100
    ra, rb = a.env.obs_rms, b.env.obs_rms
    assert np.array_equal(ra.mean, rb.mean) and np.array_equal(ra.var, rb.var) and np.array_equal(ra.count, rb.count)
    assert np.array_equal(a.returns, b.returns) and np.array_equal(a.return_rms.var, b.return_rms.var)
    a.close(), b.close()


def test_device_rollout_subenv_normalizer_on_trajectory_tensors():
    """DeviceRollout.make_subenv_normalizer(): a 16-step trajectory (with its terminal rows) normalised in ONE call per tensor equals
    sixteen single-step calls and the oracle on the same tensors, bit for bit."""
    import torch

    from gym_amd.rollout import DeviceRollout
    from oracle.oracle import SubEnvNorm

    n, K = 3000, 16
    r = DeviceRollout("Pendulum-v1", n, seed=3, action_seed=4, max_episode_steps=5)
    nz, nz1 = r.make_subenv_normalizer(gamma=0.9), r.make_subenv_normalizer(gamma=0.9)
    obs0 = r.reset(seed=3)
    with torch.cuda.stream(r.stream):
        y0 = nz.normalize_reset_obs(obs0)
        nz1.normalize_reset_obs(obs0)
        out = r.rollout_per_step(K, out=r.trajectory_buffers(K, want_final=True))
        y, yf = nz.normalize_obs(out["obs"][:K], out["final_obs"][:K], out["terminated"][:K], out["truncated"][:K])
        rw = nz.normalize_rewards(out["reward"][:K], out["terminated"][:K], out["truncated"][:K])
        ys, rs = [], []
        for k in range(K):
            yk, _ = nz1.normalize_obs(out["obs"][k], out["final_obs"][k], out["terminated"][k], out["truncated"][k])
            ys.append(yk)
            rs.append(nz1.normalize_rewards(out["reward"][k], out["terminated"][k], out["truncated"][k]))
    r.synchronize()
    assert torch.equal(y, torch.stack(ys)) and torch.equal(rw, torch.stack(rs))
    oo, orw = SubEnvNorm(r.O, n), SubEnvNorm(1, n)
    h = lambda t: t.cpu().numpy()
    z0 = np.zeros((1, n, r.O), np.float32)
    oo.observations(1, h(obs0), None, None, None, z0, True, None, 1e-8)
    assert np.array_equal(z0[0], h(y0))
    Y, YF, R = np.zeros((K, n, r.O), np.float32), np.zeros((K, n, r.O)), np.zeros((K, n))
    oo.observations(K, h(out["obs"][:K]), h(out["final_obs"][:K]), h(out["terminated"][:K]), h(out["truncated"][:K]), Y, True, YF, 1e-8)
    orw.rewards(K, h(out["reward"][:K]), False, h(out["terminated"][:K]), h(out["truncated"][:K]), R, 0.9, 1e-8)
    done = (h(out["terminated"][:K]) | h(out["truncated"][:K])).astype(bool)
    assert done.sum() >= 3 * n - n and np.array_equal(h(y), Y) and np.array_equal(h(rw), R) and np.array_equal(h(yf)[done], YF[done])
    assert np.array_equal(nz.obs_rms.var, oo.var) and np.array_equal(nz.returns, orw.returns)
    nz.close(), nz1.close(), r.close()
