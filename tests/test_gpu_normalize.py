"""SURVEY.md §8(f)-2 on the MI355X: NormalizeObservation / NormalizeReward as mxv_norm_* kernels (through the C ABI).

Parity chain (the RNG-free analogue of the step tests):
  (a) reference wrappers  == oracle mode 0, bit-exact            (tests/test_normalize_oracle.py, CPU)
  (b) device sums         == exact sums to 1e-14                 (fp64 trees vs long double)
  (c) device apply        == oracle apply GIVEN the device's sums, BIT-EXACT: everything after the sums is IEEE arithmetic
                             in the reference's order (division, sqrt, float32 roundings of the batch moments)
  (d) device end to end   vs the reference's goldens within the reference's own float32 accumulation error
                             (helpers.norm_obs_bound), rewards to rtol 1e-12
plus size-independent properties at 2^20 envs: chunk invariance (K batches at once == one at a time), invariance under
power-of-two sharding, bit-reproducibility, normalised batch moments, float32 output == rounded float64 output.
"""
import numpy as np
import pytest

from helpers import GYM_IDS, NORM_CASES, load_norm_golden, norm_obs_bound

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _oracle(g, n=None, O=None):
    from oracle.oracle import RunningNorm

    T1, gn, gO = g["raw_obs"].shape
    return RunningNorm(n or gn, O or gO, gamma=float(g["gamma"]), obs_epsilon=float(g["obs_epsilon"]),
                       rew_epsilon=float(g["rew_epsilon"]), mode=1)


@pytest.mark.parametrize("name", NORM_CASES)
def test_observations_sums_and_apply_against_oracle_and_reference(name):
    import torch
    from gym_amd import _native

    g = load_norm_golden(name)
    x = g["raw_obs"]
    K, n, O = x.shape
    nm = _native.Norm(O, n)
    xd = _dev(x)
    sums = torch.zeros((K, 2 * O), dtype=torch.float64, device="cuda")
    nm.obs_sums(K, xd, sums)
    torch.cuda.synchronize()
    orc = _oracle(g)
    np.testing.assert_allclose(sums.cpu().numpy(), orc.obs_sums(x), rtol=1e-14, atol=1e-300)        # (b)
    y = torch.empty((K, n, O), dtype=torch.float64, device="cuda")
    nm.obs_apply(K, xd, y, False, float(g["obs_epsilon"]), sums, 1, n)
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    y_orc = orc.obs_apply(x, sums.cpu().numpy()[None], n)
    assert np.array_equal(y, y_orc)                                                                    # (c)
    mean, var, count = nm.get_state()
    assert np.array_equal(mean, orc.obs_mean) and np.array_equal(var, orc.obs_var) and count == orc.obs_count[0]
    err = np.abs(y - g["nrm_obs"])
    bound = norm_obs_bound(x, g["nrm_obs"])
    assert np.all(err <= bound), float((err / bound).max())                                           # (d)
    np.testing.assert_allclose(count, float(g["obs_count"]), rtol=0, atol=0)
    nm.close()


@pytest.mark.parametrize("name", NORM_CASES)
def test_rewards_sums_and_apply_against_oracle_and_reference(name):
    import torch
    from gym_amd import _native

    g = load_norm_golden(name)
    r, te, tr = g["raw_rew"], g["term"].astype(np.uint8), g["trunc"].astype(np.uint8)
    K, n = r.shape
    gamma, eps = float(g["gamma"]), float(g["rew_epsilon"])
    nm = _native.Norm(1, n)
    rd, ted, trd = _dev(r), _dev(te), _dev(tr)
    sums = torch.zeros((K, 2), dtype=torch.float64, device="cuda")
    nm.reward_sums(K, rd, False, ted, trd, gamma, sums)
    out = torch.empty((K, n), dtype=torch.float64, device="cuda")
    nm.reward_apply(K, rd, False, out, eps, sums, 1, n)
    torch.cuda.synchronize()
    orc = _oracle(g)
    s_orc = orc.reward_sums(r, te, tr)
    np.testing.assert_allclose(sums.cpu().numpy(), s_orc, rtol=1e-14, atol=1e-300)
    o_orc = orc.reward_apply(r, sums.cpu().numpy()[None], n)
    assert np.array_equal(out.cpu().numpy(), o_orc)
    mean, var, count, returns = nm.get_state(want_returns=True)
    assert np.array_equal(returns, orc.returns) and np.array_equal(returns, g["returns"])   # the recurrence is IEEE-exact
    assert mean[0] == orc.ret_mean[0] and var[0] == orc.ret_var[0]
    np.testing.assert_allclose(out.cpu().numpy(), g["nrm_rew"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(var[0], float(g["ret_var"]), rtol=1e-12)
    nm.close()


@pytest.mark.parametrize("name", ["CartPole", "Pendulum"])
def test_one_call_forms_chunking_and_float32_output(name):
    """mxv_norm_observations / mxv_norm_rewards: K batches in one call == one batch per call (what the wrappers do), and
    the float32 output is the rounded float64 output."""
    import torch
    from gym_amd import _native

    g = load_norm_golden(name)
    x, r = g["raw_obs"], g["raw_rew"]
    te, tr = g["term"].astype(np.uint8), g["trunc"].astype(np.uint8)
    K, n, O = x.shape
    xd, rd, ted, trd = _dev(x), _dev(r), _dev(te), _dev(tr)
    a, b, c = _native.Norm(O, n), _native.Norm(O, n), _native.Norm(O, n)
    ya = torch.empty((K, n, O), dtype=torch.float64, device="cuda")
    yb = torch.empty_like(ya)
    yc = torch.empty((K, n, O), dtype=torch.float32, device="cuda")
    a.observations(K, xd, ya, False, 1e-8)
    for k in range(K):
        b.observations(1, xd[k], yb[k], False, 1e-8)
    c.observations(K, xd, yc, True, 1e-8)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    assert torch.equal(yc, ya.to(torch.float32))
    assert all(np.array_equal(u, v) for u, v in zip(a.get_state()[:2], b.get_state()[:2]))
    ra, rb = _native.Norm(1, n), _native.Norm(1, n)
    oa = torch.empty((K - 1, n), dtype=torch.float64, device="cuda")
    ob = torch.empty_like(oa)
    ra.rewards(K - 1, rd, False, ted, trd, oa, 0.99, 1e-8)
    for k in range(K - 1):
        rb.rewards(1, rd[k], False, ted[k], trd[k], ob[k], 0.99, 1e-8)
    torch.cuda.synchronize()
    assert torch.equal(oa, ob)
    # float32 rewards in, float32 out
    rf = _native.Norm(1, n)
    of = torch.empty((K - 1, n), dtype=torch.float32, device="cuda")
    rf.rewards(K - 1, rd.to(torch.float32), True, ted, trd, of, 0.99, 1e-8)
    torch.cuda.synchronize()
    np.testing.assert_allclose(of.cpu().numpy(), oa.cpu().numpy(), rtol=2e-7)
    for h in (a, b, c, ra, rb, rf):
        h.close()


@pytest.mark.parametrize("n,T", [(96, 60), (70_000, 14), (70_003, 14), (1022, 14)])   # whole + ragged leaves; n % 4 != 0 (element-wise loads), n % 4 == 2
@pytest.mark.parametrize("name", ["CartPole", "Pendulum"])
def test_wrappers_on_hip_vector_env(name, n, T):
    """gym_amd.wrappers.NormalizeReward(NormalizeObservation(HipVectorEnv)) — the reference's stacking — against the
    oracle applied to a bare twin env's outputs; attributes and dtypes of the reference's wrappers.  Both sizes read the
    step's outputs where they still sit in device-visible memory (mxv_staging_view: the pinned block of a small env, the
    device staging of a large one) and return views of pooled pinned arrays: results the caller keeps must stay intact."""
    import gym_amd
    from gym_amd.wrappers import NormalizeObservation, NormalizeReward
    from oracle.oracle import RunningNorm

    bare = gym_amd.make(GYM_IDS[name], n)
    wrapped = NormalizeReward(NormalizeObservation(gym_amd.make(GYM_IDS[name], n)), gamma=0.95)
    assert wrapped._staged and wrapped.env._staged
    kept = []
    assert wrapped.is_vector_env and wrapped.num_envs == n and wrapped.gamma == 0.95 and wrapped.epsilon == 1e-8
    bare.action_space.seed(3)
    o_b, _ = bare.reset(seed=21)
    o_w, _ = wrapped.reset(seed=21)
    O = o_b.shape[1]
    orc = RunningNorm(n, O, gamma=0.95, mode=1)
    assert o_w.dtype == np.float64 and o_w.shape == o_b.shape
    np.testing.assert_allclose(o_w, orc.normalize_obs(o_b), rtol=1e-9, atol=1e-12)
    ndone = 0
    for _ in range(T):
        a = bare.action_space.sample()
        o_b, r_b, te_b, tr_b, _ = bare.step(a)
        o_w, r_w, te_w, tr_w, _ = wrapped.step(a)
        assert np.array_equal(te_b, te_w) and np.array_equal(tr_b, tr_w)
        assert o_w.dtype == np.float64 and r_w.dtype == np.float64
        np.testing.assert_allclose(o_w, orc.normalize_obs(o_b), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r_w, orc.normalize_rewards(r_b, te_b, tr_b), rtol=1e-12, atol=0)
        ndone += int((te_b | tr_b).sum())
        if len(kept) < 4:
            kept.append((o_w, o_w.copy(), r_w, r_w.copy()))
    for a1, a2, b1, b2 in kept:
        assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    rms = wrapped.env.obs_rms
    np.testing.assert_allclose(rms.mean, orc.obs_mean, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rms.var, orc.obs_var, rtol=1e-9)
    assert rms.count == orc.obs_count[0] and abs(rms.count - n * (T + 1)) < 1e-3 * max(1, n // 96)
    np.testing.assert_allclose(wrapped.return_rms.var, orc.ret_var[0], rtol=1e-12)
    assert np.array_equal(wrapped.returns, orc.returns)
    if name == "CartPole":
        assert ndone > 0 and (wrapped.returns == 0).any()   # finished envs restart their discounted return at 0
    wrapped.close()
    bare.close()


def test_wrapper_stack_pickles_with_its_statistics():
    """NormalizeReward(NormalizeObservation(RecordEpisodeStatistics(env))) pickles as one object; the copy's running
    statistics and outputs continue identically."""
    import pickle

    import gym_amd

    env = gym_amd.make("CartPole-v1", num_envs=64, max_episode_steps=15)
    env = gym_amd.NormalizeReward(gym_amd.NormalizeObservation(gym_amd.RecordEpisodeStatistics(env)), gamma=0.95)
    env.reset(seed=4)
    env.action_space.seed(1)
    for _ in range(25):
        env.step(env.action_space.sample())
    twin = pickle.loads(pickle.dumps(env))
    assert np.array_equal(twin.obs_rms.mean, env.obs_rms.mean) and twin.obs_rms.count == env.obs_rms.count
    assert twin.return_rms.var == env.return_rms.var and np.array_equal(twin.returns, env.returns)
    for _ in range(40):
        a = env.action_space.sample()
        r0, r1 = env.step(a), twin.step(a)
        for x, y in zip(r0[:4], r1[:4]):
            assert np.array_equal(x, y)
    assert np.array_equal(twin.obs_rms.var, env.obs_rms.var) and twin.return_rms.mean == env.return_rms.mean
    env.close()
    twin.close()


def test_full_size_properties():
    """2^20 CartPole envs, 8-step trajectory chunk produced by the fused rollout kernel."""
    import torch
    from gym_amd import _native
    from gym_amd.normalize import RunningNormalizer
    from gym_amd.rollout import DeviceRollout

    n, K, O = 1 << 20, 8, 4
    dr = DeviceRollout("CartPole-v1", n, seed=4, action_seed=5)
    dr.reset(seed=4)
    tr = dr.rollout_per_step(K)
    dr.synchronize()
    x, r, te, tc = tr["obs"], tr["reward"], tr["terminated"], tr["truncated"]

    def run(chunks, world=1):
        """normalise the K batches in `chunks` calls, the env axis split over `world` emulated ranks"""
        nl = n // world
        obs_h = [_native.Norm(O, nl) for _ in range(world)]
        rew_h = [_native.Norm(1, nl) for _ in range(world)]
        y = torch.empty((K, n, O), dtype=torch.float64, device="cuda")
        o = torch.empty((K, n), dtype=torch.float64, device="cuda")
        kk = K // chunks
        for c in range(chunks):
            sl = slice(c * kk, (c + 1) * kk)
            xs = [x[sl, w * nl:(w + 1) * nl].contiguous() for w in range(world)]
            rs = [r[sl, w * nl:(w + 1) * nl].contiguous() for w in range(world)]
            tes = [te[sl, w * nl:(w + 1) * nl].contiguous() for w in range(world)]
            tcs = [tc[sl, w * nl:(w + 1) * nl].contiguous() for w in range(world)]
            s_obs = torch.zeros((world, kk, 2 * O), dtype=torch.float64, device="cuda")
            s_rew = torch.zeros((world, kk, 2), dtype=torch.float64, device="cuda")
            for w in range(world):
                obs_h[w].obs_sums(kk, xs[w], s_obs[w])
                rew_h[w].reward_sums(kk, rs[w], False, tes[w], tcs[w], 0.99, s_rew[w])
            for w in range(world):
                yw = torch.empty((kk, nl, O), dtype=torch.float64, device="cuda")
                ow = torch.empty((kk, nl), dtype=torch.float64, device="cuda")
                obs_h[w].obs_apply(kk, xs[w], yw, False, 1e-8, s_obs, world, n)
                rew_h[w].reward_apply(kk, rs[w], False, ow, 1e-8, s_rew, world, n)
                y[sl, w * nl:(w + 1) * nl] = yw
                o[sl, w * nl:(w + 1) * nl] = ow
        torch.cuda.synchronize()
        st = (obs_h[0].get_state(), rew_h[0].get_state())
        for h in obs_h + rew_h:
            h.close()
        return y, o, st

    y1, o1, st1 = run(1)
    y1b, o1b, _ = run(1)
    assert torch.equal(y1, y1b) and torch.equal(o1, o1b)                      # bit-reproducible (no atomics)
    y8, o8, st8 = run(8)
    assert torch.equal(y1, y8) and torch.equal(o1, o8)                        # chunk invariance
    for w in (2, 8):
        yw, ow, stw = run(2, world=w)
        assert torch.equal(y1, yw) and torch.equal(o1, ow), w                 # power-of-two shard invariance
        assert np.array_equal(stw[0][0], st1[0][0]) and np.array_equal(stw[0][1], st1[0][1])
        assert stw[1][1][0] == st1[1][1][0]
    # the first batch is normalised with statistics that are (up to count = 1e-4) its own moments
    y0 = y1[0]
    assert float(y0.mean(dim=0).abs().max()) < 1e-5
    assert float((y0.var(dim=0, unbiased=False) - 1).abs().max()) < 1e-4
    # the product front-end on the same tensors == the raw ABI
    rn = RunningNormalizer(n, O, stream=dr.stream)
    yp = rn.normalize_obs(x)
    op = rn.normalize_rewards(r, te, tc)
    dr.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(yp, y1) and torch.equal(op, o1)
    assert abs(rn.obs_rms.count - K * n) < 1e-3
    # returns recurrence against NumPy on a slice
    ret = np.zeros(4096)
    rr, dd = r[:, :4096].cpu().numpy(), (te | tc)[:, :4096].cpu().numpy().astype(bool)
    for k in range(K):
        ret = ret * 0.99 + rr[k]
        ret[dd[k]] = 0.0
    assert np.array_equal(rn.returns[:4096], ret)
    rn.close()
    dr.close()


def test_errors():
    from gym_amd import _native

    with pytest.raises(_native.MxvError) as ei:
        _native.Norm(5, 16)
    assert ei.value.code == _native.ERR_UNSUPPORTED
    with pytest.raises(_native.MxvError):
        _native.Norm(4, 0)
    nm = _native.Norm(4, 16)
    import torch
    x = torch.zeros((1, 16, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(_native.MxvError) as ei:
        nm.rewards(1, x, False, x, x, x, 0.99, 1e-8)          # reward statistics need dim == 1
    assert ei.value.code == _native.ERR_INVALID_ARG
    with pytest.raises(_native.MxvError):
        nm.observations(0, x, x, True, 1e-8)
    nm.close()


def test_reference_known_answers():
    """tests/wrappers/test_normalize.py of the reference, restated on the tensors its DummyRewardEnv produces: two sub-envs
    whose observation and reward at step t are (t, t+1); `obs_rms.mean` after reset / first step (:84-103) and
    `return_rms.mean` after the first / second step (:106-127), to the reference's 4 decimals."""
    import torch
    from gym_amd.normalize import RunningNormalizer

    rn = RunningNormalizer(2, 1, gamma=0.99)
    dev = torch.device("cuda")
    obs = lambda t: torch.tensor([[t], [t + 1]], dtype=torch.float32, device=dev)
    rn.normalize_obs(obs(0))                                   # envs.reset()
    np.testing.assert_almost_equal(rn.obs_rms.mean, np.mean([0.5]), decimal=4)
    rn.normalize_obs(obs(1))                                   # first step
    np.testing.assert_almost_equal(rn.obs_rms.mean, np.mean([1.0]), decimal=4)
    zeros = torch.zeros(2, dtype=torch.uint8, device=dev)
    rew = lambda t: torch.tensor([t, t + 1], dtype=torch.float64, device=dev)
    rn.normalize_rewards(rew(1), zeros, zeros)
    np.testing.assert_almost_equal(rn.return_rms.mean, np.mean([1.5]), decimal=4)
    rn.normalize_rewards(rew(2), zeros, zeros)
    np.testing.assert_almost_equal(rn.return_rms.mean, np.mean([[1, 2], [2 + rn.gamma * 1, 3 + rn.gamma * 2]]), decimal=4)
    rn.close()


def test_normalize_wrappers_fall_back_when_an_inner_wrapper_altered_the_arrays():
    """NormalizeObservation over NormalizeObservation: the outer one's input is no longer what the engine staged, so it uploads
    the array it was given (the reference's data flow) — same numbers as normalising the inner wrapper's output by hand."""
    import gym_amd
    from gym_amd.wrappers import NormalizeObservation
    from oracle.oracle import RunningNorm

    n = 50_000
    inner = NormalizeObservation(gym_amd.make("CartPole-v1", n))
    outer = NormalizeObservation(inner)
    assert inner._staged and not outer._staged
    bare = gym_amd.make("CartPole-v1", n)
    o1, o2 = RunningNorm(n, 4, mode=1), RunningNorm(n, 4, mode=1)
    ob, _ = bare.reset(seed=9)
    ow, _ = outer.reset(seed=9)
    np.testing.assert_allclose(ow, o2.normalize_obs(o1.normalize_obs(ob).astype(np.float32)), rtol=1e-6, atol=1e-9)
    bare.action_space.seed(1)
    for _ in range(5):
        a = bare.action_space.sample()
        ob = bare.step(a)[0]
        ow = outer.step(a)[0]
        np.testing.assert_allclose(ow, o2.normalize_obs(o1.normalize_obs(ob).astype(np.float32)), rtol=1e-6, atol=1e-9)
    outer.close(), bare.close()


def test_public_normalize_works_on_the_array_it_is_given():
    """NormalizeObservation.normalize(obs) is a public method of the reference (normalize.py:90-93): called directly — with an array
    that is NOT the last host step's output — it must fold and return THAT array; only step()/reset() read the staged copy."""
    import gym_amd
    from gym_amd.wrappers import NormalizeObservation
    from oracle.oracle import RunningNorm

    n = 96
    env = NormalizeObservation(gym_amd.make("CartPole-v1", n))
    bare = gym_amd.make("CartPole-v1", n)
    assert env._staged
    o0, _ = env.reset(seed=5)
    b0, _ = bare.reset(seed=5)
    rn = RunningNorm(n, 4, mode=1)
    np.testing.assert_allclose(o0, rn.normalize_obs(b0), rtol=1e-12, atol=1e-12)
    other = (np.random.default_rng(0).standard_normal((n, 4)) * 3 + 1).astype(np.float32)     # nothing the env ever produced
    got = env.normalize(other)
    np.testing.assert_allclose(got, rn.normalize_obs(other), rtol=1e-12, atol=1e-12)
    assert got.dtype == np.float64 and abs(got.mean()) < 0.6 and not np.allclose(got, o0)
    a = np.zeros(n, dtype=np.int64)
    o1, *_ = env.step(a)
    b1, *_ = bare.step(a)
    np.testing.assert_allclose(o1, rn.normalize_obs(b1), rtol=1e-12, atol=1e-12)               # the statistics include `other`
    env.close()
    bare.close()


def test_reward_normalisation_of_views_that_start_anywhere():
    """The return-sum and reward-map kernels use 16 / 32-byte vector loads where the tensors allow it; a caller's VIEW that starts one
    element into an allocation (8-byte aligned rewards, odd flag addresses) must take the element-wise path and give the same bits."""
    import torch
    from gym_amd import _native

    n, K = 8192, 12
    g = torch.Generator(device="cuda").manual_seed(3)
    rew = torch.rand((K, n), generator=g, device="cuda", dtype=torch.float64)
    te = (torch.rand((K, n), generator=g, device="cuda") < 0.05).to(torch.uint8)
    tr = (torch.rand((K, n), generator=g, device="cuda") < 0.02).to(torch.uint8)

    def shifted(t):
        flat = torch.empty(t.numel() + 1, dtype=t.dtype, device="cuda")
        v = flat[1:].view(t.shape)
        v.copy_(t)
        return v

    outs = []
    for views in (False, True):
        r, a, b = (shifted(rew), shifted(te), shifted(tr)) if views else (rew, te, tr)
        if views:
            assert r.data_ptr() % 16 == 8 and a.data_ptr() % 4 == 1
        nm = _native.Norm(1, n)
        out = shifted(torch.empty_like(rew)) if views else torch.empty_like(rew)
        nm.rewards(K, r, False, a, b, out, 0.99, 1e-8)
        torch.cuda.synchronize()
        outs.append((out.cpu().numpy().copy(), nm.get_state(want_returns=True)))
        nm.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
