""""The observation carries the state" (mxv_adopt_obs, DeviceRollout(obs_carries_state=True); round 6): single steps keep the fp64 state as
(float32 observation, int32 residual) pairs — exact by construction, so EVERYTHING must equal the ordinary path bit for bit: every step's
outputs, the fp64 state read back, the oracle; through resets, fused rollouts in between, injected states with values the pair cannot
hold (they escape to the fp64 array), and the kinds that cannot adopt say so."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = ["CartPole-v1", "MountainCar-v0", "MountainCarContinuous-v0"]


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        nan = np.isnan(a)
        return np.array_equal(nan, np.isnan(b)) and np.array_equal(a[~nan], b[~nan]) and np.array_equal(np.signbit(a[~nan]), np.signbit(b[~nan]))
    return np.array_equal(a, b)


@pytest.mark.parametrize("gid", KINDS)
@pytest.mark.parametrize("n", [1000, 70001])
def test_adopted_steps_equal_ordinary_steps_bit_for_bit(gid, n):
    import torch

    from gym_amd.rollout import DeviceRollout

    a = DeviceRollout(gid, n, seed=5, action_seed=6, max_episode_steps=17)
    b = DeviceRollout(gid, n, seed=5, action_seed=6, max_episode_steps=17, obs_carries_state=True)
    assert torch.equal(a.reset(seed=5), b.reset(seed=5))
    ended = 0
    for t in range(120):
        a.sample_actions()
        a.synchronize()                      # (the engines launch on their own streams: the sampled actions must have landed before anybody copies them)
        act = a.actions.clone()
        torch.cuda.synchronize()
        oa, ob = a.step(act), b.step(act)
        a.synchronize(), b.synchronize()
        for x, y in zip(oa, ob):
            assert torch.equal(x, y), (gid, t)
        assert torch.equal(a.final_obs, b.final_obs)
        ended += int((oa[2] | oa[3]).sum())
        if t == 0:
            assert b.handle.last_launch()["out_mode"] == 3 and a.handle.last_launch()["out_mode"] != 3
        if t == 40:      # a fused rollout in between: the state goes back to fp64 by itself, and into the pairs again at the next step
            ra, rb = a.rollout_per_step(8), b.rollout_per_step(8)
            a.synchronize(), b.synchronize()     # (compared on the default stream)
            assert all(torch.equal(ra[k], rb[k]) for k in ra)
        if t == 80:      # a partial reset
            mask = (torch.arange(n, device="cuda") % 3 == 0).to(torch.uint8)
            torch.cuda.synchronize()         # (made on the default stream, read on the engines')
            assert torch.equal(a.reset(mask=mask), b.reset(mask=mask))
    assert ended > n // 4
    sa, sb = a.handle.get_state(), b.handle.get_state()
    assert _same(sa[0], sb[0]) and np.array_equal(sa[1], sb[1]) and np.array_equal(a.handle.get_episodes(), b.handle.get_episodes())
    a.close(), b.close()


def test_one_env_per_lane_launches_equal_the_oracle_and_the_adopted_form():
    """From 2^19 envs a single CartPole step runs ONE env per lane (two rounds of waves, profiles/r6/r6i_step_launch_shape.md), in the
    ordinary and in the adopted form: the instantiation asserted, both forms bit-equal to each other over resets, and the ordinary one held
    against the oracle twin on the whole batch (masks exact, observations within the engine's ulp bar, the twin resynchronised every step)."""
    import torch

    from gym_amd.rollout import DeviceRollout
    from helpers import MAX_OBS_ULPS, ulps32
    from oracle.oracle import OracleVecEnv

    n = (1 << 19) + 77
    a = DeviceRollout("CartPole-v1", n, seed=21, action_seed=22, max_episode_steps=30)
    b = DeviceRollout("CartPole-v1", n, seed=21, action_seed=22, max_episode_steps=30, obs_carries_state=True)
    o = OracleVecEnv(0, n, 30, seed=21, action_seed=22)
    ob0 = o.reset(seed=21)
    assert np.array_equal(a.reset(seed=21).cpu().numpy(), ob0) and np.array_equal(b.reset(seed=21).cpu().numpy(), ob0)
    ended = 0
    for t in range(48):
        act = o.sample_actions()
        dev = torch.from_numpy(act).cuda()
        torch.cuda.synchronize()
        oa, obb = a.step(dev), b.step(dev)
        a.synchronize(), b.synchronize()
        for x, y in zip(oa, obb):
            assert torch.equal(x, y), t
        assert torch.equal(a.final_obs, b.final_obs)
        la, lb = a.handle.last_launch(), b.handle.last_launch()
        assert la["envs_per_lane"] == 1 and lb["envs_per_lane"] == 1 and lb["out_mode"] == 3 and la["out_mode"] != 3, (la, lb)
        ro, rr, rte, rtr, _, _ = o.step(act)
        assert np.array_equal(oa[2].cpu().numpy().astype(bool), rte) and np.array_equal(oa[3].cpu().numpy().astype(bool), rtr), t
        assert ulps32(oa[0].cpu().numpy(), ro).max() <= MAX_OBS_ULPS and np.array_equal(oa[1].cpu().numpy(), rr), t
        ended += int((rte | rtr).sum())
        if t % 8 == 7:      # the twin continues from the device's state (fp64 rounding differences must not accumulate into a mask flip)
            st, el = a.handle.get_state()
            np.testing.assert_allclose(st, o.state, rtol=1e-12, atol=1e-13)
            o.state[:] = st
    assert ended > n
    sa, sb = a.handle.get_state(), b.handle.get_state()
    assert _same(sa[0], sb[0]) and np.array_equal(sa[1], sb[1]) and np.array_equal(a.handle.get_episodes(), b.handle.get_episodes())
    a.close(), b.close()


@pytest.mark.parametrize("gid", KINDS)
def test_values_the_pair_cannot_hold_escape_and_come_back_exactly(gid):
    """Injected states with components outside float32's normal range (1e-300, 1e200), zeros of both signs, exact float32 values, NaN
    and infinities: one step in each form and the states read back must agree in every bit (NaN for NaN)."""
    import torch

    from gym_amd.rollout import DeviceRollout

    n = 4096
    rng = np.random.default_rng(0)
    a = DeviceRollout(gid, n, seed=1, action_seed=2, autoreset=False)
    b = DeviceRollout(gid, n, seed=1, action_seed=2, autoreset=False, obs_carries_state=True)
    a.reset(seed=1), b.reset(seed=1)
    st = a.handle.get_state()[0].copy()
    S = st.shape[0]
    special = [1e-300, -1e-300, 0.0, -0.0, 1e200, -1e200, np.inf, -np.inf, np.nan, 0.5, np.float64(np.float32(0.1)), 2.0 ** -140, 3e-39]
    for i, v in enumerate(special * 8):
        st[rng.integers(S), rng.integers(n)] = v
    for h in (a.handle, b.handle):
        h.set_state(st, np.full(n, 3, np.int32))
    with np.errstate(all="ignore"):
        for t in range(3):
            a.sample_actions()
            a.synchronize()
            act = a.actions.clone()
            torch.cuda.synchronize()
            oa, ob = a.step(act), b.step(act)
            a.synchronize(), b.synchronize()
            for x, y in zip(oa, ob):
                assert _same(x.cpu().numpy(), y.cpu().numpy()), (gid, t)
            assert _same(a.handle.get_state()[0], b.handle.get_state()[0]), (gid, t)      # (get_state joins: the next step splits again)
    a.close(), b.close()


def test_the_adopted_form_against_the_oracle_and_the_kinds_that_cannot_adopt():
    import torch

    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout
    from oracle.oracle import OracleVecEnv

    n = 2048
    r = DeviceRollout("CartPole-v1", n, seed=11, action_seed=12, obs_carries_state=True)
    o = OracleVecEnv(0, n, 500, seed=11, action_seed=12)
    assert np.array_equal(r.reset(seed=11).cpu().numpy(), o.reset(seed=11))
    for t in range(150):
        act = o.sample_actions()
        obs, rew, term, trunc = r.step(torch.from_numpy(act).cuda())
        r.synchronize()
        ro, rr, rte, rtr, _, _ = o.step(act)
        assert np.array_equal(term.cpu().numpy().astype(bool), rte) and np.array_equal(trunc.cpu().numpy().astype(bool), rtr), t
        np.testing.assert_allclose(obs.cpu().numpy(), ro, rtol=1e-5, atol=0)
        assert np.array_equal(rew.cpu().numpy(), rr)
        if t % 25 == 24:
            st, el = r.handle.get_state()
            np.testing.assert_allclose(st, o.state, rtol=1e-12, atol=1e-13)
            o.state[:] = st
    r.close()
    for gid in ("Pendulum-v1", "Acrobot-v1"):
        with pytest.raises(_native.MxvError, match="float32"):
            DeviceRollout(gid, 64, obs_carries_state=True)
    # a step with another observation buffer, changed physics, or a release: the ordinary path, same bits as a handle that never adopted
    a = DeviceRollout("MountainCar-v0", 512, seed=3, action_seed=4)
    b = DeviceRollout("MountainCar-v0", 512, seed=3, action_seed=4, obs_carries_state=True)
    a.reset(seed=3), b.reset(seed=3)
    for t in range(30):
        a.sample_actions()
        a.synchronize()
        act = a.actions.clone()
        torch.cuda.synchronize()
        if t == 10:
            b.handle.adopt_obs(None)
        if t == 20:
            b.handle.adopt_obs(b.obs)
        oa, ob = a.step(act), b.step(act)
        a.synchronize(), b.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(oa, ob)), t
        assert (b.handle.last_launch()["out_mode"] == 3) == (t < 10 or t >= 20)
    a.close(), b.close()
