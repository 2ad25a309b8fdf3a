"""NormalizeObservation's batch moments formed by the rollout that writes the observations (mxv_set_obs_partials; SURVEY.md §8(f)-2:
"a natural fused epilogue").  The partials [K][tiles][2 O] a sampled trajectory launch leaves behind, folded by the normaliser's tree,
are the column sums / sums of squares of the observation tensor it wrote — against float64 sums of that tensor (rtol 1e-14: a
different summation order than the stand-alone pass, both exact to fp64 rounding) for all five env kinds and both dtype sets; the
normalised observations and running statistics they lead to against the stand-alone path; bit-identical for 1 / 2 / 4 shards; and a
launch that cannot produce them says so instead of skipping them."""
import numpy as np
import pytest

from helpers import ENV_NAMES, GYM_IDS, LIMITS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("name", ENV_NAMES)
def test_partials_are_the_sums_of_the_observations_written(name, compact):
    import torch
    from gym_amd.rollout import DeviceRollout

    n, K = 33_000, 40          # whole and ragged tiles
    r = DeviceRollout(GYM_IDS[name], n, seed=2, action_seed=3, reward_f32=compact, action_i32=compact, max_episode_steps=min(LIMITS[name], 25))
    r.reset(seed=2)
    out = r.trajectory_buffers(K, layout="separate", obs_partials=True)
    leaves, per, vals = r.handle.obs_partials_layout()
    assert out["obs_partials"].shape == (K, leaves, vals) and vals == 2 * r.O and leaves == -(-n // per)
    twin = DeviceRollout(GYM_IDS[name], n, seed=2, action_seed=3, reward_f32=compact, action_i32=compact, max_episode_steps=min(LIMITS[name], 25))
    twin.reset(seed=2)
    ref = twin.trajectory_buffers(K, layout="separate")
    for rep in range(2):
        r.rollout_per_step(K, out=out)
        twin.rollout_per_step(K, out=ref)
        r.synchronize(), twin.synchronize()
        li = r.handle.last_launch()
        assert li["kernel"] == 1 and li["out_mode"] == (2 if compact else 1) and li["safe"] == 0
        for key in ("obs", "reward", "terminated", "truncated", "actions"):      # the STATS instantiation changes no output
            assert torch.equal(out[key], ref[key]), (name, key)
        x = out["obs"].cpu().numpy().astype(np.float64)                          # [K, n, O]
        p = out["obs_partials"].cpu().numpy()
        O = r.O
        pad = leaves * per - n
        xp = np.concatenate([x, np.zeros((K, pad, O))], axis=1).reshape(K, leaves, per, O)
        np.testing.assert_allclose(p[:, :, :O], xp.sum(axis=2), rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(p[:, :, O:], (xp * xp).sum(axis=2), rtol=1e-13, atol=1e-300)
        assert int((out["terminated"] | out["truncated"]).sum()) > 0
    r.close(), twin.close()


def test_normalised_observations_through_the_fused_moments():
    import torch
    from gym_amd.rollout import DeviceRollout

    n, K = 1 << 18, 24
    r = DeviceRollout("CartPole-v1", n, seed=5, action_seed=6)
    r.reset(seed=5)
    out = r.trajectory_buffers(K, layout="separate", obs_partials=True)
    a, b = r.make_normalizer(), r.make_normalizer()
    for rep in range(3):
        r.rollout_per_step(K, out=out)
        ya = a.normalize_obs(out["obs"], partials=out["obs_partials"])
        yb = b.normalize_obs(out["obs"])
        r.synchronize()
        np.testing.assert_allclose(ya.cpu().numpy(), yb.cpu().numpy(), rtol=1e-9, atol=1e-12)
    for u, v in zip(a.backend.obs_state(), b.backend.obs_state()):
        np.testing.assert_allclose(u, v, rtol=1e-12)
    r.close()


def test_partial_sums_do_not_depend_on_the_sharding():
    """One handle of 2^18 envs, two of 2^17, four of 2^16 (global env indices): the tree's [K][2 O] sums are bit-identical — tiles are
    the same 128 envs whatever the shard size (the STATS launches never switch to one env per lane)."""
    import torch
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    n, K = 1 << 18, 12
    results = []
    for shards in (1, 2, 4):
        m = n // shards
        per_shard = []
        for w in range(shards):
            r = DeviceRollout("CartPole-v1", m, env_offset=w * m, seed=7, action_seed=8)
            r.reset(seed=7)
            out = r.trajectory_buffers(K, layout="separate", obs_partials=True)
            r.rollout_per_step(K, out=out)
            r.synchronize()
            assert r.handle.last_launch()["envs_per_lane"] == 2
            nm = _native.Norm(4, m, stream=r.stream.cuda_stream)
            sums = torch.empty((K, 8), dtype=torch.float64, device="cuda")
            nm.obs_sums_partials(K, out["obs_partials"], out["obs_partials"].shape[1], sums)
            r.synchronize()
            per_shard.append(sums.cpu().numpy())
            nm.close(), r.close()
        tot = per_shard
        while len(tot) > 1:                                  # the binary tree over ranks (scan_kernel's order)
            tot = [tot[i] + tot[i + 1] for i in range(0, len(tot), 2)]
        results.append(tot[0])
    assert np.array_equal(results[0], results[1]) and np.array_equal(results[0], results[2])


def test_a_launch_that_cannot_produce_them_fails_loudly():
    import torch
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout("CartPole-v1", 4096, seed=1, action_seed=2)
    r.reset(seed=1)
    out = r.trajectory_buffers(8, layout="separate", obs_partials=True)
    p = r.handle.get_params()
    p[0] = 9.0                                              # gravity: runtime parameters -> no STATS instantiation
    r.handle.set_params(p)
    with pytest.raises(_native.MxvError) as e:
        r.rollout_per_step(8, out=out)
    assert e.value.code == _native.ERR_UNSUPPORTED and "mxv_set_obs_partials" in str(e.value)
    plain = {k: v for k, v in out.items() if k != "obs_partials"}
    r.rollout_per_step(8, out=plain)                         # detached: the ordinary launch
    r.synchronize()
    r.close()


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("name,both", [("CartPole", True), ("CartPole", False), ("Pendulum", True), ("MountainCarContinuous", False), ("Acrobot", True)])
def test_discounted_returns_advanced_by_the_rollout(name, both, compact):
    """NormalizeReward fused the same way (mxv_set_return_partials): the rollout advances the normaliser's running returns and leaves
    their per-step sums; normalised rewards, running statistics and the returns array equal the stand-alone path's (the recurrence is
    IEEE-exact, so the returns are bit-identical; the sums differ in summation order only)."""
    import torch
    from gym_amd.rollout import DeviceRollout

    n, K = 70_000, 32
    kw = dict(seed=4, action_seed=5, reward_f32=compact, action_i32=compact, max_episode_steps=min(LIMITS[name], 21))
    a, b = DeviceRollout(GYM_IDS[name], n, **kw), DeviceRollout(GYM_IDS[name], n, **kw)
    a.reset(seed=4), b.reset(seed=4)
    na, nb = a.make_normalizer(gamma=0.97), b.make_normalizer(gamma=0.97)
    a.fuse_reward_normalizer(na)
    oa = a.trajectory_buffers(K, layout="separate", obs_partials=both, ret_partials=True)
    ob = b.trajectory_buffers(K, layout="separate")
    for rep in range(3):
        a.rollout_per_step(K, out=oa)
        b.rollout_per_step(K, out=ob)
        ra = na.normalize_rewards(oa["reward"], oa["terminated"], oa["truncated"], partials=oa["ret_partials"])
        rb = nb.normalize_rewards(ob["reward"], ob["terminated"], ob["truncated"])
        a.synchronize(), b.synchronize()
        for key in ("obs", "reward", "terminated", "truncated", "actions"):
            assert torch.equal(oa[key], ob[key]), (name, key)
        np.testing.assert_allclose(ra.cpu().numpy(), rb.cpu().numpy(), rtol=(2e-6 if compact else 1e-11), atol=0)
        if both:
            ya = na.normalize_obs(oa["obs"], partials=oa["obs_partials"])
            yb = nb.normalize_obs(ob["obs"])
            a.synchronize(), b.synchronize()
            np.testing.assert_allclose(ya.cpu().numpy(), yb.cpu().numpy(), rtol=1e-9, atol=1e-12)
    sa, sb = na.backend.reward_state(), nb.backend.reward_state()
    assert np.array_equal(sa[3], sb[3]) and (sa[3] != 0).any()       # the running returns themselves: bit-identical
    assert int((oa["terminated"] | oa["truncated"]).sum()) > 0
    np.testing.assert_allclose(sa[1], sb[1], rtol=1e-12)
    assert sa[2] == sb[2]
    li = a.handle.last_launch()
    assert li["kernel"] == 1 and li["out_mode"] == (2 if compact else 1)
    a.close(), b.close()


def test_attached_partials_never_go_stale_silently():
    """C ABI: with a partials buffer attached, every launch that does not fill it fails (single steps, tape launches, eager rollouts,
    final-tensor rollouts); the Python front-end detaches by itself when a call shape without partials follows."""
    import torch
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout("CartPole-v1", 2048, seed=1, action_seed=2)
    r.reset(seed=1)
    out = r.trajectory_buffers(4, layout="separate", obs_partials=True)
    r.rollout_per_step(4, out=out)
    r.synchronize()
    h = r.handle
    dev_a = torch.zeros(2048, dtype=torch.int64, device="cuda")
    for call in (lambda: h.step(dev_a.data_ptr(), r.obs, r.reward, r.terminated, r.truncated, None),
                 lambda: h.rollout(4, r.obs, r.reward, r.terminated, r.truncated, None, None, per_step=False, mode=_native.ROLLOUT_FUSED),
                 lambda: h.rollout(4, out["obs"], out["reward"], out["terminated"], out["truncated"], None, out["actions"], per_step=True, mode=_native.ROLLOUT_EAGER),
                 lambda: h.rollout_tape(4, out["actions"], out["obs"], out["reward"], out["terminated"], out["truncated"], None, per_step=True)):
        with pytest.raises(_native.MxvError) as e:
            call()
        assert e.value.code == _native.ERR_UNSUPPORTED
    # the front-end: other call shapes detach first
    r.step(dev_a)
    r.rollout(3)
    plain = {k: v for k, v in out.items() if k != "obs_partials"}
    r.rollout_tape(out["actions"].clone(), out=plain)
    with pytest.raises(ValueError):
        r.rollout_tape(out["actions"].clone(), out=out)
    r.rollout_per_step(4, out=out)                      # and re-attach
    r.synchronize()
    assert torch.isfinite(out["obs_partials"]).all()
    r.close()


@pytest.mark.parametrize("name,n", [("CartPole", 1), ("CartPole", 130), ("Pendulum", 65), ("Acrobot", 3), ("MountainCar", 257)])
def test_partials_of_batches_smaller_than_a_tile(name, n):
    """One env, one env more than a tile: the lanes without an env contribute zeros; both moment sets; odd launch lengths."""
    import torch
    from gym_amd.rollout import DeviceRollout

    r = DeviceRollout(GYM_IDS[name], n, seed=8, action_seed=9, max_episode_steps=min(LIMITS[name], 7))
    r.reset(seed=8)
    nz = r.make_normalizer(gamma=0.9)
    r.fuse_reward_normalizer(nz)
    leaves, per, vals = r.handle.obs_partials_layout()
    ret = np.zeros(n)
    for K in (2, 5, 3):
        out = r.trajectory_buffers(K, layout="separate", obs_partials=True, ret_partials=True)
        r.rollout_per_step(K, out=out)
        r.synchronize()
        x = out["obs"].cpu().numpy().astype(np.float64)
        pad = leaves * per - n
        xp = np.concatenate([x, np.zeros((K, pad, r.O))], axis=1).reshape(K, leaves, per, r.O)
        p = out["obs_partials"].cpu().numpy()
        np.testing.assert_allclose(p[:, :, :r.O], xp.sum(axis=2), rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(p[:, :, r.O:], (xp * xp).sum(axis=2), rtol=1e-13, atol=1e-300)
        rew = out["reward"].cpu().numpy().astype(np.float64)
        done = (out["terminated"] | out["truncated"]).cpu().numpy().astype(bool)
        rp = out["ret_partials"].cpu().numpy()
        for k in range(K):                                    # normalize.py:132-136 restated
            ret = ret * 0.9 + rew[k]
            rpad = np.concatenate([ret, np.zeros(pad)]).reshape(leaves, per)
            np.testing.assert_allclose(rp[k, :, 0], rpad.sum(axis=1), rtol=1e-13, atol=1e-300)
            np.testing.assert_allclose(rp[k, :, 1], (rpad * rpad).sum(axis=1), rtol=1e-13, atol=1e-300)
            ret[done[k]] = 0.0
    assert np.array_equal(nz.backend.reward_state()[3], ret)
    r.close()
