"""NormalizeObservation / NormalizeReward over the trajectory tensors, stand-alone and with the moments fused into the rollout."""
from .common import *  # noqa: F401,F403
from .common import _event_us, _hbm, _spin


def measure_normalize(torch, envs, chunk, reps=6):
    """NormalizeObservation / NormalizeReward (gym/wrappers/normalize.py:57-144) on the [K][N] trajectory tensors of a
    fused CartPole rollout: per chunk, the batch moments of every step (one streaming read), then the affine map with the statistics as
    they stood after that step's update (read + write).  Algorithmic bytes per env-step: observations 4 O (sums) + 4 O + 4 O (apply,
    float32 out) = 48; rewards 8 + 2 (sums: reward + both flags) + 8 + 8 (apply) = 26."""
    from gym_amd import _native
    from gym_amd.rollout import DeviceRollout

    dr = DeviceRollout(ENV_ID, envs, seed=0, action_seed=1)
    dr.reset(seed=0)
    tr = dr.rollout_per_step(chunk, out=dr.trajectory_buffers(chunk, layout="separate"))
    dr.synchronize()
    s, O = dr.stream, dr.O
    no, nr = _native.Norm(O, envs, stream=s.cuda_stream), _native.Norm(1, envs, stream=s.cuda_stream)
    with torch.cuda.stream(s):
        y32 = torch.empty((chunk, envs, O), dtype=torch.float32, device=dr.device)
        o64 = torch.empty((chunk, envs), dtype=torch.float64, device=dr.device)
    out = {"workload": f"{ENV_ID}, num_envs={envs}, the [K={chunk}][N] trajectory tensors of one fused launch normalised in place of the "
                       "reference's per-step wrappers (running mean / var updated once per step, exactly their order)",
           "normalize_obs": _hbm(_event_us(torch, s, lambda: no.observations(chunk, tr["obs"], y32, True, 1e-8), reps, chunk), envs, 12 * O,
                                 kernels="mxv_norm.hip: obs sums (read 4 O) + scan + apply (read 4 O, write 4 O float32)"),
           "normalize_reward": _hbm(_event_us(torch, s, lambda: nr.rewards(chunk, tr["reward"], False, tr["terminated"], tr["truncated"], o64, 0.99, 1e-8),
                                              reps, chunk), envs, 26, kernels="mxv_norm.hip: discounted-return sums (read 8 + 2) + scan + apply (read 8, write 8)")}
    # the batch moments formed by the rollout itself (mxv_set_obs_partials): what NormalizeObservation then costs ON TOP of the rollout
    try:
        tr = None                                                  # (the set normalised above: its numbers are taken, its 9 GiB are needed)
        torch.cuda.empty_cache()
        trp = dr.trajectory_buffers(chunk, obs_partials=True)     # sorted by HBM class, as a caller gets them by default
        plain = {k: t for k, t in trp.items() if k != "obs_partials"}
        nf = _native.Norm(O, envs, stream=s.cuda_stream)
        r0 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=plain), reps, chunk)
        r1 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=trp), reps, chunk)
        sums = torch.empty((chunk, 2 * O), dtype=torch.float64, device=dr.device)

        def fused():
            nf.obs_sums_partials(chunk, trp["obs_partials"], trp["obs_partials"].shape[1], sums)
            nf.obs_apply(chunk, trp["obs"], y32, True, 1e-8, sums.unsqueeze(0), 1, envs)

        nfu = _event_us(torch, s, fused, reps, chunk)
        inc = nfu + (r1 - r0)
        b = 8 * O + 2 * 16 * O * trp["obs_partials"].shape[1] / envs   # apply: read 4 O + write 4 O; partials: 2 O doubles per tile, written + read
        out["normalize_obs_fused_moments"] = dict(_hbm(inc, envs, b, kernels="rollout_kernel_v3<..., STATS> writes per-tile column sums; mxv_norm.hip: tree over the "
                                                                              "partials + scan + apply (read 4 O, write 4 O float32); no pass reads the observations back"),
                                                  rollout_us_per_step=r0, rollout_with_partials_us_per_step=r1, normalize_from_partials_us_per_step=nfu,
                                                  separate_us_per_step=out["normalize_obs"]["us_per_step"],
                                                  note="us_per_step = what normalisation adds to the rollout: (rollout with partials - rollout) + tree + scan + apply")
        # ... and NormalizeReward's discounted returns (mxv_set_return_partials), alone and together with the observation moments
        nrf = _native.Norm(1, envs, stream=s.cuda_stream)

        class _Returns:      # what DeviceRollout.fuse_reward_normalizer needs of a normaliser: its returns array and its discount
            gamma = 0.99
            backend = nrf

        dr.fuse_reward_normalizer(_Returns)
        leaves = trp["obs_partials"].shape[1]
        with torch.cuda.stream(s):
            rp = torch.empty((chunk, leaves, 2), dtype=torch.float64, device=dr.device)
        rets, both = dict(plain, ret_partials=rp), dict(trp, ret_partials=rp)
        r2 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=rets), reps, chunk)
        r3 = _event_us(torch, s, lambda: dr.rollout_per_step(chunk, out=both), reps, chunk)
        rsums = torch.empty((chunk, 2), dtype=torch.float64, device=dr.device)

        def fused_reward():
            nrf.reward_sums_partials(chunk, rp, leaves, rsums)
            nrf.reward_apply(chunk, trp["reward"], False, o64, 1e-8, rsums.unsqueeze(0), 1, envs)

        nru = _event_us(torch, s, fused_reward, reps, chunk)
        out["normalize_reward_fused_moments"] = dict(_hbm(nru + (r2 - r0), envs, 16 + 2 * 16 * leaves / envs,
                                                          kernels="rollout_kernel_v3<..., STATS = 2> advances the discounted returns; tree + scan + apply (read 8, write 8)"),
                                                     rollout_with_partials_us_per_step=r2, normalize_from_partials_us_per_step=nru,
                                                     separate_us_per_step=out["normalize_reward"]["us_per_step"])
        out["rollout_and_both_normalisations"] = {"separate_us_per_step": r0 + out["normalize_obs"]["us_per_step"] + out["normalize_reward"]["us_per_step"],
                                                  "fused_us_per_step": r3 + nfu + nru, "rollout_with_both_partials_us_per_step": r3}
        dr.handle.set_obs_partials(None)
        dr.handle.set_return_partials(None, 0.0, None)            # nothing may point into nrf's returns any more
        nf.close(), nrf.close()
        del trp, plain, sums, rp, rets, both, rsums
    except Exception as e:  # noqa: BLE001
        out.setdefault("normalize_obs_fused_moments", {"error": f"{type(e).__name__}: {e}"[:300]})
        out.setdefault("normalize_reward_fused_moments", {"error": f"{type(e).__name__}: {e}"[:300]})
    no.close(), nr.close(), dr.close()
    del tr, y32, o64
    torch.cuda.empty_cache()
    return out


def measure_subenv_normalize(torch, envs, chunk=64, reps=6):
    """The PER-SUB-ENV NormalizeObservation / NormalizeReward of `make(wrappers=[...])` (gym/vector/__init__.py:56-65 around
    gym/wrappers/normalize.py:50-145: every sub-env its own RunningMeanStd, batches of one) at `envs` sub-envs, three ways:
    the mxv_subnorm_* kernels on a [K][N] trajectory (statistics in registers across the K steps: 8 O + 2 algorithmic bytes per
    env-step for the observations, 8 + 8 + 2 for the rewards), the same kernels called once per step (K = 1: + 16 (2 O + 1) bytes of
    statistics each way), and the host-side NumPy statistics the wrappers used before round 6 (and still use below 4096 sub-envs)."""
    import time

    import numpy as np

    from gym_amd.rollout import DeviceRollout
    from gym_amd.wrappers import _PerEnvMeanStd

    dr = DeviceRollout(ENV_ID, envs, seed=0, action_seed=1)
    obs0 = dr.reset(seed=0)
    tr = dr.rollout_per_step(chunk, out=dr.trajectory_buffers(chunk, want_final=True, layout="separate"))
    dr.synchronize()
    s, O = dr.stream, dr.O
    nz = dr.make_subenv_normalizer()
    with torch.cuda.stream(s):
        nz.normalize_reset_obs(obs0)
        y = torch.empty((chunk, envs, O), dtype=torch.float32, device=dr.device)
        yf = torch.empty((chunk, envs, O), dtype=torch.float64, device=dr.device)
        ro = torch.empty_like(tr["reward"])
    args = (tr["obs"], tr["final_obs"], tr["terminated"], tr["truncated"])
    obs_k = _event_us(torch, s, lambda: nz.normalize_obs(*args, out=y, final_out=yf), reps, chunk)
    rew_k = _event_us(torch, s, lambda: nz.normalize_rewards(tr["reward"], tr["terminated"], tr["truncated"], out=ro), reps, chunk)
    one = lambda k: nz.normalize_obs(tr["obs"][k], tr["final_obs"][k], tr["terminated"][k], tr["truncated"][k], out=y[k], final_out=yf[k])  # noqa: E731
    obs_1 = _event_us(torch, s, lambda: [one(k) for k in range(chunk)], reps, chunk)
    rew_1 = _event_us(torch, s, lambda: [nz.normalize_rewards(tr["reward"][k], tr["terminated"][k], tr["truncated"][k], out=ro[k]) for k in range(chunk)],
                      reps, chunk)
    # the host statistics on the same step (NumPy over [N][O] float64 arrays; inputs already on the host: PCIe not counted)
    x = tr["obs"][0].cpu().numpy()
    h = _PerEnvMeanStd(envs, (O,))
    h.update(x)
    t0 = time.perf_counter()
    for _ in range(3):
        h.update(x)
        _ = ((x - h.mean) / np.sqrt(h.var + 1e-8)).astype(np.float32)
    host_us = (time.perf_counter() - t0) / 3 * 1e6
    out = {"workload": f"{ENV_ID}, num_envs={envs}: per-sub-env running statistics (batches of one row), K = {chunk} trajectory steps per call "
                       "vs one call per step vs the host NumPy statistics",
           "normalize_obs": _hbm(obs_k, envs, 8 * O + 2, kernels="mxv_subnorm.hip: subnorm_obs_kernel (read 4 O + 2 flag bytes, write 4 O float32; "
                                                                    "terminal rows of finished sub-envs: + 4 O read, 8 O written, ~5 % of rows)"),
           "normalize_reward": _hbm(rew_k, envs, 18, kernels="mxv_subnorm.hip: subnorm_rew_kernel (read 8 + 2, write 8)"),
           "normalize_obs_one_call_per_step": _hbm(obs_1, envs, 8 * O + 2 + 16 * (2 * O + 1), kernels="the same kernel, K = 1: the statistics make the round trip"),
           "normalize_reward_one_call_per_step": _hbm(rew_1, envs, 18 + 16 * 4, kernels="K = 1"),
           "host_numpy_obs_us_per_step": host_us, "device_over_host": host_us / obs_1}
    nz.close(), dr.close()
    del tr, y, yf, ro
    torch.cuda.empty_cache()
    return out
